// dpq_cli.cpp -- `deltapq`, the reference-compatible command line driver for the
// query path, on top of the C-ABI (include/deltapq_amd.h).
//
// Mirrors the flag surface and output of the reference driver
// (/root/reference/deltapq_approx_tree_main.cpp:14-70 flags, :265-349 `-task
// query`, :617-710 `-task query_im`, :72-149 `-task approx_tree` with -method 1):
//
//   deltapq -dataset DIR -task query -m 8 -k 256 -h 1 -diff 8 -N 1000000
//           -query_size 1000 -topk 100 [-ext fvecs|bvecs] [-debug]
//           [-gpus G] [-out FILE]
//
// Files read from DIR, same names as the reference: M{m}K{k}codewords.txt
// (main:274), query.{ext} (main:303), M{m}K{k}_Approx_compressed_codes_opt_N{N}
// (h:2812-2814).  -h, -diff and -method are parsed and ignored by the query, as
// in the reference.  Both `query` and `query_im` load the
// index once into HBM (the reference re-opens the file per query for `query`).
// Extensions: -gpus G shards the index over G GPUs of this node and merges the
// partial top-k lists on the host; -out writes all results (the reference
// only prints top-1 under -debug); -task pqscan is the reference's uncompressed
// comparator (main:496-556); -task encode is the encode step of the reference's
// other binary, pqtree (main.cpp:314-425), so that base vectors -> codes ->
// index -> query runs from this one tool.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>
#include <thread>
#include <vector>

#include "../../include/deltapq_amd.h"

static double Elapsed() {  // utils.cpp:112-116
    using namespace std::chrono;
    return duration<double>(system_clock::now().time_since_epoch()).count();
}

static int die(const char* where, int rc) {
    std::cout << where << ": " << dpq_strerror(rc) << ": " << dpq_last_error() << std::endl;
    return 1;
}

int main(int argc, char* argv[]) {
    std::string dataset, queryset, task = "approx_tree", ext = "fvecs", out_path;
    int query_size = -1, top_k = 1, diff_argument = 1, debug = 0, max_height_folds = 1, method = 1;
    int PQ_M = 0, PQ_K = 0, gpus = 1;
    long long N = -1;
    for (int i = 0; i < argc; i++) {  // main:26-70: hand-rolled scan, no validation
        std::string arg = argv[i];
        const char* nx = i + 1 < argc ? argv[i + 1] : "";
        if (arg == "-dataset") dataset = nx;
        if (arg == "-queryset") queryset = nx;
        if (arg == "-task") task = nx;
        if (arg == "-topk") top_k = atoi(nx);
        if (arg == "-N") N = atoll(nx);
        if (arg == "-diff") diff_argument = atoi(nx);
        if (arg == "-query_size") query_size = atoi(nx);
        if (arg == "-m") PQ_M = atoi(nx);
        if (arg == "-k") PQ_K = atoi(nx);
        if (arg == "-ext") ext = nx;
        if (arg == "-debug") debug = 1;
        if (arg == "-h") max_height_folds = atoi(nx);  // max height folds, not help (main:61-63)
        if (arg == "-method") method = atoi(nx);
        if (arg == "-gpus") gpus = atoi(nx);
        if (arg == "-out") out_path = nx;
    }
    (void)diff_argument; (void)method; (void)queryset;

    if (task == "encode") {
        // The other binary's `pqtree -task encode` (main.cpp:314-425): base.{ext} -> PQ codes (nearest
        // codeword per sub-space, PQTree::EncodePlain pq_tree.cpp:215-237) -> codes.bin.plain.M{M}K{K}N{N}.
        if (PQ_M <= 0 || PQ_K <= 0 || dataset.empty()) {
            std::cout << "usage: deltapq -dataset DIR -task encode -m M -k K [-N N] [-ext fvecs|bvecs]" << std::endl;
            return 2;
        }
        const std::string cw_path =
            dataset + "/M" + std::to_string(PQ_M) + "K" + std::to_string(PQ_K) + "codewords.txt";
        int32_t cM = 0, cK = 0, cDs = 0;
        int rc = dpq_read_codewords(cw_path.c_str(), &cM, &cK, &cDs, nullptr);
        if (rc) return die("ReadCodewords", rc);
        std::vector<float> codewords((size_t)cM * cK * cDs);
        rc = dpq_read_codewords(cw_path.c_str(), &cM, &cK, &cDs, codewords.data());
        if (rc) return die("ReadCodewords", rc);
        if (cM != PQ_M || cK != PQ_K) {
            std::cout << "codewords file is M=" << cM << " K=" << cK << std::endl;
            return 1;
        }
        const std::string base_path = dataset + "/base." + ext;  // main.cpp:354
        int64_t n_file = 0;
        int32_t D = 0;
        rc = dpq_read_vecs(base_path.c_str(), ext == "bvecs", &n_file, &D, nullptr, 0);
        if (rc) return die("ItrReader", rc);
        int64_t n = n_file;
        if (N != -1 && N < n) n = N;  // main.cpp:339-340: -N caps the number of vectors
        std::vector<float> base((size_t)n * D);
        rc = dpq_read_vecs(base_path.c_str(), ext == "bvecs", &n_file, &D, base.data(), n);
        if (rc) return die("ItrReader", rc);
        const double t0 = Elapsed();
        std::vector<uint8_t> codes((size_t)n * PQ_M);
        rc = dpq_encode_pq(base.data(), n, D, codewords.data(), PQ_M, PQ_K, cDs, 0, codes.data());
        if (rc) return die("encode", rc);
        const std::string out = dataset + "/codes.bin.plain.M" + std::to_string(PQ_M) + "K" + std::to_string(PQ_K) +
                                "N" + std::to_string(n);  // main.cpp:409-411
        rc = dpq_write_codes_plain(out.c_str(), codes.data(), n, PQ_M);
        if (rc) return die("PQTree::Write", rc);
        std::cout << "N = " << n << std::endl;                                         // pq_tree.cpp:1021
        std::cout << "encoded " << n << " vectors in " << (Elapsed() - t0) << " [sec] -> " << out << std::endl;
        return 0;
    }
    if (task == "approx_tree") {
        // main:72-149: codes.bin.plain -> DeltaTree -> the three index artefacts
        if (PQ_M <= 0 || PQ_K <= 0 || dataset.empty() || N < 0) {
            std::cout << "usage: deltapq -dataset DIR -task approx_tree -m M -k K -N N [-h FOLDS]" << std::endl;
            return 2;
        }
        std::cout << "M = " << PQ_M << std::endl;
        const std::string codes_path = dataset + "/codes.bin.plain.M" + std::to_string(PQ_M) + "K" +
                                       std::to_string(PQ_K) + "N" + std::to_string(N);  // main:76-77
        int64_t NN = 0;
        int rc = dpq_read_codes_plain(codes_path.c_str(), PQ_M, &NN, nullptr);
        if (rc) return die("PQTree::Read", rc);
        std::cout << "Read: N = " << NN << std::endl;
        std::vector<uint8_t> vecs((size_t)NN * PQ_M);
        rc = dpq_read_codes_plain(codes_path.c_str(), PQ_M, &NN, vecs.data());
        if (rc) return die("PQTree::Read", rc);
        const std::string cw_path =
            dataset + "/M" + std::to_string(PQ_M) + "K" + std::to_string(PQ_K) + "codewords.txt";  // main:88-89
        int32_t cM = 0, cK = 0, cDs = 0;
        rc = dpq_read_codewords(cw_path.c_str(), &cM, &cK, &cDs, nullptr);
        if (rc) return die("ReadCodewords", rc);
        std::vector<float> codewords((size_t)cM * cK * cDs);
        rc = dpq_read_codewords(cw_path.c_str(), &cM, &cK, &cDs, codewords.data());
        if (rc) return die("ReadCodewords", rc);
        if (cM != PQ_M || cK != PQ_K) {
            std::cout << "codewords file is M=" << cM << " K=" << cK << std::endl;
            return 1;
        }
        std::cout << "K = " << PQ_K << std::endl << "N = " << NN << std::endl << dataset << std::endl;
        const double t0 = Elapsed();  // main:98
        dpq_tree* tree = nullptr;
        // the sort/group passes run on GPU 0 when there is one (same tree either way); -cpu_build forces the host
        bool cpu_build = dpq_device_count() < 1;
        for (int i = 0; i < argc; i++)
            if (std::string(argv[i]) == "-cpu_build") cpu_build = true;
        std::cout << "edge search on " << (cpu_build ? "the host" : "GPU 0") << std::endl;
        rc = cpu_build ? dpq_tree_build(vecs.data(), NN, PQ_M, PQ_K, max_height_folds, codewords.data(), cDs, &tree)
                       : dpq_tree_build_gpu(vecs.data(), NN, PQ_M, PQ_K, max_height_folds, codewords.data(), cDs, 0,
                                            &tree);
        if (rc) return die("create_approx_tree", rc);
        dpq_dtc_stats st;
        dpq_tree_stats(tree, &st);
        std::cout << "   ++++ TOTAL number of Diffs " << st.n_diffs << std::endl;                       // h:1315
        for (int d = 0; d < PQ_M + 2 && d < 16; ++d) std::cout << st.depth_hist[d] << " nodes at depth " << d << std::endl;  // h:1467-1469
        std::cout << "number of bytes is " << st.n_bytes << std::endl;                                  // h:1768
        rc = dpq_tree_write_files(tree, dataset.c_str());
        if (rc) return die("write index files", rc);
        dpq_tree_free(tree);
        std::cout << "==========================BUILD DELTATREE INDEX IN " << (Elapsed() - t0) << " [sec] "
                  << "==========================" << std::endl << std::endl;                           // main:136-137
        std::cout << "WARNING: Just built an index. no query processed." << std::endl;                 // main:140
        return 0;
    }
    const bool pqscan = task == "pqscan";  // main:496-556: uncompressed comparator over codes.bin.plain
    // -task batch_query (main:351-420) is accepted as an alias of -task query: the engine behind `query`
    // already decodes every chunk once for a whole batch of queries, with the -task query arithmetic and ids
    // (the reference's batch variant accumulates in fp32 and records the second node of a pair under the
    // first one's id, h:3079, h:3389-3392 -- not reproduced).
    if (task == "batch_query") {
        std::cout << "NOTE: -task batch_query runs as -task query here (its arithmetic, ids and tie order); the reference's "
                     "batch variant (fp32 accumulation, pair ids: h:3079, h:3389-3392) is a documented deviation, not reproduced"
                  << std::endl;
        task = "query";
    }
    if (task != "query" && task != "query_im" && !pqscan) {
        std::cout << "deltapq (MI355X build): -task query, query_im, pqscan, approx_tree and encode are implemented (batch_query = "
                     "alias of query); got '" << task
                  << "'" << std::endl;
        return 2;
    }
    if (PQ_M <= 0 || PQ_K <= 0 || dataset.empty()) {
        std::cout << "usage: deltapq -dataset DIR -task query -m M -k K -N N -query_size Q -topk K [-ext fvecs|bvecs]"
                     " [-debug] [-gpus G] [-out FILE]" << std::endl;
        return 2;
    }

    // main:274-275
    const std::string cw_path = dataset + "/M" + std::to_string(PQ_M) + "K" + std::to_string(PQ_K) + "codewords.txt";
    std::cout << cw_path << std::endl;
    int32_t cM = 0, cK = 0, cDs = 0;
    int rc = dpq_read_codewords(cw_path.c_str(), &cM, &cK, &cDs, nullptr);
    if (rc) return die("ReadCodewords", rc);
    std::vector<float> codewords((size_t)cM * cK * cDs);
    rc = dpq_read_codewords(cw_path.c_str(), &cM, &cK, &cDs, codewords.data());
    if (rc) return die("ReadCodewords", rc);
    std::cout << "++++++ codewords read from +++++" << cw_path << std::endl;
    std::cout << "++++++ " << cM << "Ks " << cK << "Ds " << cDs << std::endl;
    if (cM != PQ_M || cK != PQ_K) {
        std::cout << "codewords file is M=" << cM << " K=" << cK << " but -m " << PQ_M << " -k " << PQ_K << std::endl;
        return 1;
    }

    // index file (h:2812-2814), or the plain code file for pqscan (h:2616-2618); like the
    // reference the caller must pass -N (it is part of the file name)
    char fname[4096];
    int64_t n_codes = 0, n_bytes = 0;
    if (pqscan) {
        // the reference's pqscan opens codes.bin.plain.M{M}K{K} (h:2616-2617); the encoder and approx_tree
        // use the name with the N{N} suffix (main:76-77) -- accept either
        snprintf(fname, sizeof fname, "%s/codes.bin.plain.M%dK%d", dataset.c_str(), PQ_M, PQ_K);
        rc = dpq_read_codes_plain(fname, PQ_M, &n_codes, nullptr);
        if (rc) {
            snprintf(fname, sizeof fname, "%s/codes.bin.plain.M%dK%dN%lld", dataset.c_str(), PQ_M, PQ_K, N);
            rc = dpq_read_codes_plain(fname, PQ_M, &n_codes, nullptr);
        }
        std::cout << fname << std::endl;
        if (rc) return die("open codes", rc);
        std::cout << "top_k = " << top_k << std::endl;  // main:504
    } else {
        rc = dpq_dtc_file_name(dataset.c_str(), PQ_M, PQ_K, N, fname, sizeof fname);
        if (rc) return die("file name", rc);
        std::cout << fname << std::endl;
        rc = dpq_read_dtc_header(fname, &n_codes, &n_bytes);
        if (rc) return die("open index", rc);
    }
    if (N == -1) N = n_codes;  // h:2825
    if (N < 1 || N > n_codes) {
        std::cout << "-N " << N << " outside 1.." << n_codes << std::endl;
        return 1;
    }
    if (N != n_codes)  // h:2826-2829 / main:629-632: the first N codes of a larger index
        std::cout << "scan only part of the codes " << N << " / " << n_codes << std::endl;
    std::cout << "M = " << PQ_M << std::endl;
    std::cout << "K = " << PQ_K << std::endl;
    std::cout << "N = " << N << std::endl;
    std::cout << dataset << std::endl;

    // main:303-308
    const std::string q_path = dataset + "/query." + ext;
    std::cout << "In ReadTopN " << q_path << std::endl;
    int64_t nq_file = 0;
    int32_t D = 0;
    rc = dpq_read_vecs(q_path.c_str(), ext == "bvecs", &nq_file, &D, nullptr, 0);
    if (rc) return die("ReadTopN", rc);
    std::cout << nq_file << " query vectors read from " << q_path << std::endl;
    int64_t nq = nq_file;
    if (nq > 10000) nq = 10000;
    if (query_size != -1) {
        if (query_size > nq) {
            std::cout << "-query_size " << query_size << " exceeds the " << nq << " available queries" << std::endl;
            return 1;
        }
        nq = query_size;
    }
    std::vector<float> queries((size_t)nq * D);
    rc = dpq_read_vecs(q_path.c_str(), ext == "bvecs", &nq_file, &D, queries.data(), nq);
    if (rc) return die("ReadTopN", rc);
    if (D != PQ_M * cDs) {
        std::cout << "query dimension " << D << " != M*Ds = " << PQ_M * cDs << std::endl;
        return 1;
    }

    const int ndev = dpq_device_count();
    if (ndev < 1) {
        std::cout << "no GPU visible: this build has no CPU query path" << std::endl;
        return 1;
    }
    if (gpus < 1) gpus = 1;
    if (gpus > ndev) {
        std::cout << "-gpus " << gpus << " but only " << ndev << " device(s) visible" << std::endl;
        return 1;
    }

    // load once: parse + transcode + upload, one shard per GPU
    std::vector<dpq_index*> shards((size_t)gpus, nullptr);
    const double tl0 = Elapsed();
    for (int g = 0; g < gpus; ++g) {
        dpq_open_opts o;
        memset(&o, 0, sizeof o);
        o.device = g;
        o.shard_rank = g;
        o.shard_count = gpus;
        o.num_codes = N != n_codes ? (int32_t)N : 0;
        rc = pqscan ? dpq_open_plain_file(fname, PQ_M, PQ_K, &o, &shards[(size_t)g])
                    : dpq_open_file(fname, PQ_M, PQ_K, &o, &shards[(size_t)g]);
        if (rc) return die("dpq_open_file", rc);
        rc = dpq_set_codebook(shards[(size_t)g], codewords.data(), cDs);
        if (rc) return die("dpq_set_codebook", rc);
    }
    std::cout << "index resident on " << gpus << " GPU(s) in " << (Elapsed() - tl0) << " [sec]" << std::endl;

    // ranked_scores[q] = vector<pair<int,float>>(top_k)  (main:310-311), flattened
    std::vector<int32_t> ids((size_t)nq * top_k);
    std::vector<float> dists((size_t)nq * top_k);
    std::vector<int32_t> part_ids;
    std::vector<float> part_dists;
    if (gpus > 1) {
        part_ids.resize((size_t)gpus * nq * top_k);
        part_dists.resize((size_t)gpus * nq * top_k);
    }

    const double t0 = Elapsed();  // main:327
    if (gpus == 1) {
        // the reference's loop, one query per call (main:328-339), as batches in flight: host buffers in and out, the
        // copies of neighbouring batches beside a batch's kernels (dpq_query_batch_host_async, page-locked buffers)
        const int64_t chunk = 1024;
        const int D = PQ_M * cDs;
        dpq_pin_host(queries.data(), (int64_t)(queries.size() * sizeof(float)));   // (not fatal if refused: the copies then run unpipelined)
        dpq_pin_host(ids.data(), (int64_t)(ids.size() * sizeof(int32_t)));
        dpq_pin_host(dists.data(), (int64_t)(dists.size() * sizeof(float)));
        for (int64_t q0 = 0; q0 < nq && !rc; q0 += chunk) {
            const int n = (int)std::min<int64_t>(chunk, nq - q0);
            rc = dpq_query_batch_host_async(shards[0], queries.data() + (size_t)q0 * D, n, top_k, ids.data() + (size_t)q0 * top_k,
                                            dists.data() + (size_t)q0 * top_k);
        }
        if (rc) return die("dpq_query_batch_host_async", rc);
        rc = dpq_finish(shards[0]);
        if (rc) return die("dpq_finish", rc);
        dpq_unpin_host(queries.data());
        dpq_unpin_host(ids.data());
        dpq_unpin_host(dists.data());
    } else {
        std::vector<int> rcs((size_t)gpus, 0);
        std::vector<std::string> msgs((size_t)gpus);
        std::vector<std::thread> th;
        for (int g = 0; g < gpus; ++g)
            th.emplace_back([&, g]() {
                rcs[(size_t)g] = dpq_query_batch(shards[(size_t)g], queries.data(), (int)nq, top_k,
                                                 part_ids.data() + (size_t)g * nq * top_k,
                                                 part_dists.data() + (size_t)g * nq * top_k);
                if (rcs[(size_t)g]) msgs[(size_t)g] = dpq_last_error();
            });
        for (auto& t : th) t.join();
        for (int g = 0; g < gpus; ++g)
            if (rcs[(size_t)g]) {
                std::cout << "shard " << g << ": " << dpq_strerror(rcs[(size_t)g]) << ": " << msgs[(size_t)g]
                          << std::endl;
                return 1;
            }
        rc = dpq_merge_topk_host(part_ids.data(), part_dists.data(), gpus, (int)nq, top_k, ids.data(), dists.data());
        if (rc) return die("dpq_merge_topk_host", rc);
    }
    const double elapsed = Elapsed() - t0;
    if (debug)  // main:340-343: top-1 per query
        for (int64_t q = 0; q < nq; ++q)
            std::cout << ids[(size_t)q * top_k] << " " << dists[(size_t)q * top_k] << std::endl;
    std::cout << elapsed / (double)nq * 1000 << " [msec/query] " << std::endl;  // main:345
    std::cout << nq << " queries run" << std::endl;                              // main:347

    if (!out_path.empty()) {
        FILE* f = fopen(out_path.c_str(), "wb");
        if (!f) {
            std::cout << "cannot open " << out_path << std::endl;
            return 1;
        }
        int64_t hdr[2] = {nq, top_k};
        fwrite(hdr, sizeof(int64_t), 2, f);
        fwrite(ids.data(), sizeof(int32_t), ids.size(), f);
        fwrite(dists.data(), sizeof(float), dists.size(), f);
        fclose(f);
    }
    for (auto* s : shards) dpq_close(s);
    std::cout << "===========================" << std::endl << std::endl;  // main:711
    return 0;
}
