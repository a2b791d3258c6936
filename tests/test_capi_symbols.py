"""The C-ABI library loads without a GPU and exports exactly what
include/deltapq_amd.h declares (no compute calls here)."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    text = open(os.path.join(ROOT, "include", "deltapq_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dpq_[a-z_0-9]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    from deltapq_amd import _lib
    declared = header_functions()
    bound = sorted(name for name, _, _ in _lib.SYMBOLS)
    assert declared == bound, "header and ctypes binding disagree"
    raw = ctypes.CDLL(_lib.LIB_PATH)        # the HIP runtime is already bound by the `lib` fixture
    for name in declared:
        assert hasattr(raw, name), name
    # ... and nothing else: every dpq_* the library exports is declared in the header (developer diagnostics included)
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted({line.split()[-1] for line in out.splitlines() if line.split() and line.split()[-1].startswith("dpq_")
                       and line.split()[-2] in ("T", "t", "W")})
    assert exported == declared, sorted(set(exported) ^ set(declared))
    assert lib.dpq_version() == 100
    assert lib.dpq_strerror(0) == b"ok" and lib.dpq_strerror(-3) == b"malformed DTC stream"


def test_struct_layouts_match_header(lib):
    from deltapq_amd import _lib
    assert ctypes.sizeof(_lib.OpenOpts) == 48 + 8 * 4 + 8
    assert ctypes.sizeof(_lib.Info) == 7 * 8 + 8 * 4 + 8 + 2 * 4 + 8
    assert ctypes.sizeof(_lib.Profile) == 3 * 8 + 10 * 8 + 2 * 8 + 2 * 8 + 3 * 8
    assert ctypes.sizeof(_lib.DtcStats) == 3 * 8 + 16 * 8 + 2 * 4


def test_oracle_is_not_linked_into_the_product(lib):
    """The product library must not depend on the oracle (no CPU fallback)."""
    from deltapq_amd import _lib
    blob = open(_lib.LIB_PATH, "rb").read()
    assert b"liboracle" not in blob and b"oracle_" not in blob
    for root, _, files in os.walk(os.path.join(ROOT, "deltapq_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                src = open(os.path.join(root, f), errors="replace").read()
                assert "import oracle" not in src and "from oracle" not in src and "dtc_oracle" not in src, f


def test_open_without_gpu_fails_loudly(lib):
    from deltapq_amd import api
    if api.device_count() > 0:
        pytest.skip("a GPU is present")
    from conftest import make_case
    tree, payload, nb = make_case(100, seed=1)
    with pytest.raises(api.DpqError) as e:
        api.DeltaPQIndex.open_memory(payload, 100)
    assert e.value.status == -4 and "no CPU fallback" in str(e.value)


def test_cli_usage_without_gpu(built):
    import subprocess
    exe = os.path.join(ROOT, "deltapq_amd", "csrc", "deltapq")
    r = subprocess.run([exe, "-task", "diff_scan"], capture_output=True, text=True)   # an out-of-scope ablation task
    assert r.returncode == 2 and "are implemented" in r.stdout
    r = subprocess.run([exe, "-task", "batch_query"], capture_output=True, text=True)  # alias of query: asks for its flags
    assert r.returncode == 2 and "usage: deltapq" in r.stdout and "documented deviation" in r.stdout


def test_cli_approx_tree_builds_the_index_without_gpu(built, tmp_path):
    """`deltapq -task approx_tree` (main:72-149) is host code: codes.bin.plain -> the three artefacts."""
    import subprocess
    from deltapq_amd import api, synth
    d = str(tmp_path)
    n = 3000
    cb = synth.make_codebook(8, 256, 16, seed=1)
    synth.write_codewords_txt(os.path.join(d, "M8K256codewords.txt"), cb)
    rng = np.random.default_rng(0)
    protos = rng.integers(0, 256, size=(40, 8), dtype=np.uint8)
    codes = protos[rng.integers(0, 40, size=n)].copy()
    codes[np.arange(n), rng.integers(0, 8, size=n)] = rng.integers(0, 256, size=n)
    api.write_codes_plain(os.path.join(d, "codes.bin.plain.M8K256N%d" % n), codes)
    exe = os.path.join(ROOT, "deltapq_amd", "csrc", "deltapq")
    r = subprocess.run([exe, "-dataset", d, "-task", "approx_tree", "-m", "8", "-k", "256", "-h", "1", "-diff", "8",
                        "-N", str(n)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "BUILD DELTATREE INDEX IN" in r.stdout and "no query processed" in r.stdout
    n_codes, payload = api.read_dtc_file(synth.dtc_file_name(d, 8, 256, n))
    assert n_codes == n
    api.dtc_validate(payload, n)
    vec_id = api.read_qnode_ids(os.path.join(d, "M8K256_Approx_TreeNodesDFS_N%d" % n), n)
    assert sorted(vec_id.tolist()) == list(range(n))


def test_developer_diagnostics_are_refused_outside_dev_mode(lib):
    """dpq_debug_* (declared in the header's developer section) answer DPQ_ERR_STATE unless the process runs with DPQ_DEV=1,
    and read no environment variable before that check."""
    import ctypes as ct
    if os.environ.get("DPQ_DEV", "0") not in ("", "0"):
        pytest.skip("DPQ_DEV is set in this environment")
    ms = ct.c_float()
    fake = ct.c_void_p(0x1000)   # never dereferenced: the mode check comes first
    assert lib.dpq_debug_scan_time(fake, 1, 0, 1, 0, ms) == -7
    assert lib.dpq_debug_select_time(fake, 1, 10, 0, 1, ms) == -7
    assert lib.dpq_debug_strand1_stamps(fake, None, 0) == -7
    assert b"DPQ_DEV=1" in lib.dpq_last_error()
