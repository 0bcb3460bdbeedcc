import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ob():
    """The oracle binding (test infrastructure only)."""
    from oracle import binding
    binding.build()
    return binding


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.exists(s) and os.path.getmtime(s) > t + 1.0 for s in sources)


@pytest.fixture(scope="session")
def pkg():
    """The product package (ctypes binding of libprach_hip.so).  Built on demand: when the library or the CLI binary is
    missing or older than a source file (neither is tracked in git)."""
    import glob
    import __graft_entry__ as g
    csrc = os.path.join(g.PKG_DIR, "csrc")
    srcs = glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h")) + glob.glob(os.path.join(csrc, "*.c")) + \
        [os.path.join(ROOT, "include", "prach.h")]
    if any(_stale(os.path.join(g.PKG_DIR, f), srcs) for f in ("libprach_hip.so", "libprach_hip_tinyq.so", "prach_sim")):
        g.build()
    return g.load_package()


@pytest.fixture(scope="session")
def engine(pkg):
    eng = pkg.Engine(0)  # raises loudly when there is no device: GPU tests must not pass on a fallback
    yield eng
    eng.close()


def load_golden(case):
    path = os.path.join(GOLDEN_DIR, f"{case}.json")
    if not os.path.exists(path):
        pytest.skip(f"fixture {case}.json not generated")
    with open(path) as f:
        return json.load(f)


def golden_cfg_kwargs(g):
    return dict(g["cfg_overrides"])


def split_stdout_blocks(text):
    """Reference stdout -> list of per-trial blocks (starting at the '-------- NNNNN Result' line)."""
    blocks, cur = [], None
    for line in text.split("\n"):
        if line.startswith("-------- "):
            if cur is not None:
                blocks.append("\n".join(cur) + "\n")
            cur = [line]
        elif cur is not None and line != "":
            cur.append(line)
    if cur is not None:
        blocks.append("\n".join(cur) + "\n")
    return blocks
