"""rtmi_film_gather with more than one rank, on one GPU: tests/comm_ranks_worker.py in a process of its own (threads as
ranks, tests/fake_rccl as the collective library).  Covers what a one-rank job cannot: peers that send while the root
receives, the status rounds across real ranks, a root that owns no pixels, a rank that changes its film, errors on one
rank reaching all of them."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fake_rccl_exports_what_the_library_binds():
    """Not a GPU test: the stand-in offers exactly the nine entry points csrc/comm.cpp looks up."""
    import re
    src = open(os.path.join(ROOT, "raytracing_amd", "csrc", "comm.cpp")).read()
    wanted = set(re.findall(r'sym\("(nccl\w+)"\)', src))
    assert len(wanted) == 9
    lib = os.path.join(ROOT, "tests", "fake_rccl", "_build", "libfake_rccl.so")
    if not os.path.exists(lib):
        subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "fake_rccl")], check=True, capture_output=True)
    out = subprocess.run(["nm", "-D", "--defined-only", lib], check=True, capture_output=True, text=True).stdout
    have = set(re.findall(r" T (nccl\w+)", out))
    assert wanted <= have, wanted - have
    # and the product never names the stand-in
    for dirpath, _, files in os.walk(os.path.join(ROOT, "raytracing_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip")):
                assert "fake_rccl" not in open(os.path.join(dirpath, f), errors="ignore").read(), f


@pytest.mark.gpu
def test_gather_between_several_ranks_on_one_gpu():
    env = dict(os.environ)
    env.pop("RTMI_RCCL_LIB", None)
    r = subprocess.run([sys.executable, "-u", os.path.join(ROOT, "tests", "comm_ranks_worker.py")], env=env,
                       capture_output=True, text=True, timeout=600)
    print(r.stdout)
    print(r.stderr[-4000:])
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-4000:]
    lines = r.stdout.splitlines()
    assert "ALL OK" in lines
    assert sum(1 for ln in lines if ln.startswith("ok ")) == 5
