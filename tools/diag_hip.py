import ctypes, os, sys
print({k:v for k,v in os.environ.items() if any(s in k for s in ("HIP","ROCR","HSA","LD_LIB","GPU","ROCM"))})
lib=ctypes.CDLL(os.path.join(os.environ.get("GRAFT_REPO_ROOT","/root/repo"),"pico_tree_amd/csrc/libptk.so"))
lib.ptk_last_error.restype=ctypes.c_char_p
print("device_count without torch:", lib.ptk_device_count())
import subprocess
print(open("/proc/self/maps").read().count("libamdhip64"))
for l in open("/proc/self/maps"):
    if "amdhip" in l or "hsa-runtime" in l:
        print(l.split()[-1]); 
