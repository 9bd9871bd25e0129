import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mode = sys.argv[1]
def maps():
    m = open("/proc/self/maps").read()
    return sorted({l.split()[-1] for l in m.splitlines() if "amdhip" in l or "hsa-runtime" in l})
if mode == "lib_first":
    from min_llm_inference_amd import load_library
    lib = load_library()
    print("after lib:", maps())
    import torch
    print("avail", torch.cuda.is_available())
    print("after torch:", maps())
else:
    import torch
    print("avail", torch.cuda.is_available())
    from min_llm_inference_amd import load_library
    lib = load_library()
    print("after both:", maps())
if mode.endswith("tensor"):
    x = torch.zeros(4, device="cuda:0")
n = ctypes.c_int(-1)
print("hipGetDeviceCount rc", lib.hipGetDeviceCount(ctypes.byref(n)), n.value)
print("hipSetDevice rc", lib.hipSetDevice(0))
y = torch.ones(4, device="cuda:0")
print(y.sum().item())
