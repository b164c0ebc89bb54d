"""Localises the round-1 GroupNorm / side-stream fault.  Run on a library built by
    tools/build_variant.sh xcheck -DGMK_GN_XCHECK
    GMK_LIBGMK=generative_models_amd/libgmk_xcheck.so python tools/gn_xcheck.py [runs]
In that build gn_silu_fwd_reg_kernel reduces its per-lane partial sums TWICE from the same registers - once with the
ds_bpermute butterfly (whose results it goes on with), once with DPP / v_permlane swaps - and records every lane where the two
disagree.  Repeated identical forwards with the 1x1 skip convolution on the side stream then tell the two hypotheses apart:
  * forwards differ AND mismatching lanes are recorded  -> the cross-lane step itself returns wrong data (not the loads, not LDS `red`)
  * forwards differ, no mismatch recorded               -> the reduction is innocent; look at the loaded x / the LDS exchange
A second, less perturbing form:  tools/build_variant.sh xcheck2 -DGMK_GN_XCHECK=2 -DGMK_SHFL_BPERMUTE  keeps the kernel as the
failing build had it and adds one DPP compare per total (lanes l and l ^ 8 must agree bit for bit after a correct butterfly);
the record then shows (this lane's totals | lane ^ 8's totals).
"""
import ctypes, struct, sys
import torch
sys.path.insert(0, ".")
from generative_models_amd import ops
ops.FWD_SIDE = True          # the overlap under investigation (off by default in the product)
from generative_models_amd._lib import lib
from generative_models_amd.diffusion.simple_unet import SimpleUnet

assert hasattr(lib, "gmk_debug_gn_xcheck"), "needs a -DGMK_GN_XCHECK build (see the docstring)"
def readout(reset=True):
    buf = (ctypes.c_uint * (16 + 8 * 16))()
    torch.cuda.synchronize()
    assert lib.gmk_debug_gn_xcheck(buf, int(reset)) == 0
    return list(buf)
f32 = lambda u: struct.unpack("f", struct.pack("I", u))[0]

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 16
torch.manual_seed(0)
net = SimpleUnet(128, 0.0).cuda().eval()
with torch.no_grad():
    for n, p in net.named_parameters():
        if ".out_layers.3.weight" in n:
            p.uniform_(-0.02, 0.02)
net.mark_params_changed()
B = 1024
z = torch.randn(B, 1, 28, 28, device="cuda"); l = torch.randn(B, device="cuda"); y = torch.randint(0, 10, (B,), device="cuda")
for side in (True, False):
    ops.WGRAD_STREAM = side
    readout()
    outs = []
    for r in range(runs):
        outs.append(net.forward_hip(z, l, y, None).clone())
        d = readout()
        if d[0]:
            print(f"  run {r}: {d[0]} mismatching lanes of {d[1]} waves checked")
            for k in range(min(d[0], 8)):
                rec = d[16 + 16 * k: 32 + 16 * k]
                print(f"    block {rec[0]} tid {rec[1]} (wave {rec[1] >> 6} lane {rec[1] & 63}) HW {rec[2]} NVEC {rec[3]}  bpermute {[f32(v) for v in rec[4:8]]}  dpp {[f32(v) for v in rec[8:12]]}  lane input {[f32(v) for v in rec[12:16]]}")
    same = [bool(torch.equal(outs[0], o)) for o in outs]
    print(f"side stream {'on ' if side else 'off'}: {same.count(True)} of {runs} forwards identical to the first")
