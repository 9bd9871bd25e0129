"""The C++ drop-in surface on a real GPU: tests/cpp/engine_gpu_test.cpp is reference-style driver code (Tensor,
layers, *InferenceModel, ItemStorage, MemoryBlockManager, start_*_engine) compiled with plain g++ against
min_llm_inference_amd/host/include and linked with libmli_hip.so.  One child process, run once."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_cpp_engines_finish_and_agree_under_both_allocation_flavours(mli):
    """The reference runs its whole suite twice, USE_ASYNC_ALLOC ON and OFF (Makefile:20-30).  Here the driver is built
    twice -- plain, and with -DDEFAULT_ALLOC_METHOD=1, which makes ASYNC_ALLOCATE the flavour of every tensor in the
    process, the library's own included -- and both runs must finish every engine with the same tokens."""
    cpp = os.path.join(HERE, "cpp")
    r = subprocess.run(["make", "-C", cpp, "gpu"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    sums = {}
    for binary, flavour in (("engine_gpu_test", "SYNC_ALLOCATE"), ("engine_gpu_test_async", "ASYNC_ALLOCATE")):
        r = subprocess.run([os.path.join(cpp, "build", binary)], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and "ALL ENGINES AGREE" in r.stdout, r.stdout[-4000:] + r.stderr[-4000:]
        assert f"allocation flavour: {flavour}" in r.stdout, r.stdout[-2000:]
        sums[flavour] = [ln for ln in r.stdout.splitlines() if ln.startswith("TOKENS CHECKSUM")][0]
    assert sums["SYNC_ALLOCATE"] == sums["ASYNC_ALLOCATE"], sums
