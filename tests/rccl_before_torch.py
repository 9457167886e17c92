"""The C ABI's RCCL entry used BEFORE PyTorch is imported (a pure-ctypes integrator that later imports torch): the process must
end cleanly (round 5: with the system librccl opened RTLD_GLOBAL it ended in "double free or corruption" inside rocm_smi)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))      # (run as a script by tests/test_gpu_multidevice.py)
L = ctypes.CDLL(os.environ.get("ROBCHAR_HIP_LIB") or os.path.join(ROOT, "code-robchar_amd", "csrc", "librobchar_hip.so"))
comm = ctypes.c_void_p()
devs = (ctypes.c_int * 1)(0)
print("rc_comm_init", L.rc_comm_init(1, devs, ctypes.byref(comm)), flush=True)
print("rc_comm_destroy", L.rc_comm_destroy(comm), flush=True)
import torch
t = torch.zeros(8, device="cuda"); s = torch.cuda.Stream(); torch.cuda.synchronize()
libs = sorted({l.split()[-1] for l in open("/proc/self/maps") if "smi" in l.lower() or "rccl" in l.lower()})
print("mapped:", libs, flush=True)
print("end", flush=True)
