"""oracle/pq_encode_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement of the reference's PQ encoder, PQTree::EncodePlain (/root/reference/pq_tree.cpp:215-237):
per sub-space the nearest codeword in fp32 -- `diff = v - c; dist += diff * diff` with separately rounded
multiply and add, strict `<` so that the first minimum wins.  PARITY UNPINNED (see dtc_oracle.cpp): written from
reading the source; the reference cannot be built here (OpenCV).  The GPU encoder (encode_pq_kernel) is compared
with this bit for bit."""
import numpy as np


def encode_pq(vectors, codebook):
    v = np.asarray(vectors, dtype=np.float32)
    cb = np.asarray(codebook, dtype=np.float32)
    M, K, Ds = cb.shape
    codes = np.zeros((len(v), M), dtype=np.uint8)
    for m in range(M):
        sub = v[:, m * Ds:(m + 1) * Ds]
        dist = np.zeros((len(v), K), dtype=np.float32)
        for d in range(Ds):
            diff = (sub[:, d:d + 1] - cb[m, :, d][None, :]).astype(np.float32)
            dist = (dist + (diff * diff).astype(np.float32)).astype(np.float32)
        codes[:, m] = dist.argmin(1)              # argmin returns the first minimum
    return codes
