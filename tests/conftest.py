import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "parallel-sequence-alignment_amd")
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """CPU oracle (test infrastructure); built with gcc on first use."""
    from oracle import oracle as orc
    orc.build()
    return orc


@pytest.fixture(scope="session")
def psa():
    """ctypes binding of libpsa_amd.so; built with hipcc if the .so is missing."""
    lib_path = os.path.join(PKG, "psa_amd", "libpsa_amd.so")
    if not os.path.exists(lib_path):
        import __graft_entry__
        __graft_entry__.build()
    # torch first: its wheel bundles its own HIP runtime, which must be the first one initialised in a
    # process that also uses torch.cuda (the tests with device-resident inputs); libpsa_amd.so links /opt/rocm's
    import torch  # noqa: F401
    import psa_amd
    return psa_amd


@pytest.fixture(scope="session")
def psa_debug(psa):
    """A second instance of the binding on libpsa_debug.so (-DPSA_EXPERIMENT): the only build that reads the
    experiment / fault-injection environment knobs (PSA_SEQ_R, PSA_DUO_FORCE_TIMEOUT, ...)."""
    import importlib.util
    lib_path = os.path.join(PKG, "psa_amd", "libpsa_debug.so")
    if not os.path.exists(lib_path):
        import __graft_entry__
        __graft_entry__.build()
    old = os.environ.get("PSA_AMD_LIB")
    os.environ["PSA_AMD_LIB"] = lib_path
    try:
        spec = importlib.util.spec_from_file_location("psa_amd_debug", os.path.join(PKG, "psa_amd", "__init__.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        if old is None:
            del os.environ["PSA_AMD_LIB"]
        else:
            os.environ["PSA_AMD_LIB"] = old
    return mod


@pytest.fixture(scope="session")
def bench_line(psa):
    """The JSON line of ONE `python bench.py --steps 2 --warmup 1 --queries 64 --no-cpu-baseline` run on this box (a child
    process; exit code 0 = its self-check passed), shared by the structural test (test_gpu_parity.py) and the thresholds
    (test_zz_bench_contract.py)."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--queries", "64",
                          "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env=env)
    assert run.returncode == 0, run.stderr[-3000:]
    lines = [ln for ln in run.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


@pytest.fixture(scope="session")
def reference_records(oracle):
    """The 15 stacked records of the reference's input.txt (tests/golden copy)."""
    return oracle.read_records(os.path.join(GOLDEN, "reference_input.txt"))


# distinct records of input.txt (SURVEY.md Appendix B numbering)
DISTINCT_RECORDS = [0, 1, 4, 5, 7, 8, 9, 10, 11, 13]
