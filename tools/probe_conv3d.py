"""Time gldm_conv3d_k3 on the shipped encoder's shapes (GLDM_LIB selects a diagnostic build of the library)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graspldm_amd import _lib as L
if os.environ.get("GLDM_LIB"):
    L.LIB_PATH = os.environ["GLDM_LIB"]
from graspldm_amd.voxel import pack_conv3d
B = 256
for cin, cout, r in [(48, 48, 24), (96, 96, 12), (48, 96, 12), (3, 48, 24)]:
    x = torch.randn(B, cin, r, r, r, device="cuda")
    w = torch.randn(cout, cin, 3, 3, 3) * 0.05
    wp = pack_conv3d(w).cuda()
    bias = torch.randn(cout, device="cuda")
    y = torch.empty(B, cout, r, r, r, device="cuda")
    part = torch.empty(int(L.lib().gldm_conv3d_partial_floats(B, cout, r)), device="cuda")
    f = lambda: L.call("gldm_conv3d_k3", L.ptr(x), L.ptr(wp), L.ptr(bias), B, cin, cout, r, L.ptr(y), L.ptr(part), L.current_stream(x.device))
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    fl = 2 * 27 * cin * cout * r ** 3 * B
    print(f"conv3d {cin:3d}->{cout:3d} @ {r}^3 x {B}: {ms:7.3f} ms  {fl / ms / 1e9:7.1f} TFLOP/s  {fl / ms / 1e9 / 157.3 * 100:5.1f} % of f32 MFMA peak")
