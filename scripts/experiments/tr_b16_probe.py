"""Runs the ds_read_b64_tr_b16 probe on the GPU box and prints the lane/element map."""
import ctypes, os, subprocess, torch
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "tr_b16_probe.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-shared", "-fPIC", os.path.join(here, "tr_b16_probe.hip"), "-o", so])
lib = ctypes.CDLL(so)
for mode in (0, 1):
    out = torch.zeros(256, dtype=torch.int16, device="cuda")
    rc = lib.run_probe(ctypes.c_void_p(out.data_ptr()), mode)
    o = out.cpu().view(64, 4).tolist()
    print("mode", mode, "rc", rc)
    for l in (0, 1, 2, 3, 4, 5, 15, 16, 17, 31, 32, 47, 48, 63):
        print("  lane", l, o[l])
