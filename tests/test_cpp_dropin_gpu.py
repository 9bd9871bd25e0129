"""The C++ drop-in surface on a real GPU: tests/cpp/engine_gpu_test.cpp is reference-style driver code (Tensor,
layers, *InferenceModel, ItemStorage, MemoryBlockManager, start_*_engine) compiled with plain g++ against
min_llm_inference_amd/host/include and linked with libmli_hip.so.  One child process, run once."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_cpp_engines_finish_and_agree(mli):
    cpp = os.path.join(HERE, "cpp")
    r = subprocess.run(["make", "-C", cpp, "gpu"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    r = subprocess.run([os.path.join(cpp, "build", "engine_gpu_test")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ALL ENGINES AGREE" in r.stdout, r.stdout[-4000:] + r.stderr[-4000:]
