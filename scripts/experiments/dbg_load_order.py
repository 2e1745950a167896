import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
order = sys.argv[1]
if order == "torch_first":
    import torch
    print("torch avail", torch.cuda.is_available(), torch.version.hip)
import divans_amd as da
L = da.load_library()
h = ctypes.c_void_p()
cfg = da.config_simple()
rc = L.divans_gpu_codec_create(ctypes.byref(h), ctypes.byref(cfg), 0, None, 4096)
print(order, "create rc", rc, L.divans_gpu_last_error())
if order != "torch_first":
    import torch
    print("torch avail", torch.cuda.is_available(), torch.version.hip)
    x = torch.zeros(4, device="cuda"); print(x.sum().item())
with open("/proc/self/maps") as f:
    libs = sorted({l.split()[-1] for l in f if "libamdhip64" in l or "libhsa-runtime" in l})
print(libs)
