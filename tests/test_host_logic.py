"""Host scheduler (continuous batching + KV page allocator) unit tests, CPU only.

The C++ sources under min_llm_inference_amd/host/src are compiled with g++ (ASan + UBSan) against a malloc
test double of the Tensor memory backend (tests/cpp/memory_host_double.cpp) and run as one binary whose
scenarios mirror the reference's tests/item_storage_test.cpp and tests/paged_item_storage_test.cpp."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def test_host_scheduler_cpp_suite():
    cpp = os.path.join(HERE, "cpp")
    r = subprocess.run(["make", "-C", cpp], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    r = subprocess.run([os.path.join(cpp, "build", "host_logic_test")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "0 failure(s)" in r.stdout, r.stdout[-4000:] + r.stderr[-4000:]
    assert r.stdout.count("[ OK ]") >= 12


def test_pipelined_loop_logic_against_sequential_loop_cpu():
    """tests/cpp/pipelined_logic_test.cpp: the pipelined engine loop with a fake forward over the malloc double, 40
    random shapes / pool sizes / EOF rates, against the reference's loop order; ASan + UBSan."""
    cpp = os.path.join(HERE, "cpp")
    r = subprocess.run(["make", "-C", cpp], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    r = subprocess.run([os.path.join(cpp, "build", "pipelined_logic_test")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "0 failure(s)" in r.stdout, r.stdout[-4000:] + r.stderr[-4000:]
    assert r.stdout.count("[ OK ]") == 141   # 70 random + 60 tight pools + 4 page-boundary EOF + 7 too-small pools
