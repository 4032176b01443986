"""Dev tool (round 5): the EAGER mean-teacher step with its branch streams confined to disjoint CU sets
(hipExtStreamCreateWithCUMask: bit b = CU b // 8 of XCC b % 8, tools/probe/cumask_graph.hip) - spatial partitioning instead of time sharing.
Phase times from HIP events on the main stream, as tools/stage_stamps.py.

usage: python tools/exp_cumask.py CONFIG [FIELD=INT ...]
  CONFIG = plain | tea:N | split:T,S,M | wg:N
    plain        three ordinary streams (the product)
    tea:N        teacher stream on CUs [0, N) of every XCC, the student streams on every CU
    teax:N       teacher stream on CUs [0, N), both student streams on CUs [N, 32)
    split:T,S    teacher on [0, T), target-domain student stream on [T, T + S), main (source pass, losses, tail) on [T + S, 32)
    wg:N         staged weight gradients (policy wgrad_overlap / wgrad_cap from FIELD=INT) on a stream confined to CUs [0, N), chains unmasked
    wgx:N        ... and both chain streams confined to CUs [N, 32)
"""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uda_poseestimation_amd import synthetic, warp
from uda_poseestimation_amd.engine import MeanTeacherTrainer
import uda_poseestimation_amd.lib.models as models

cfg = sys.argv[1] if len(sys.argv) > 1 else "plain"
pol = {k: int(v) for k, v in (a.split("=") for a in sys.argv[2:] if "=" in a)}
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
hip = C.CDLL("libamdhip64.so")


def masked_stream(lo, hi):
    """torch stream whose kernels run on CUs [lo, hi) of every XCC."""
    m = (C.c_uint32 * 8)()
    for b in range(256):
        if lo <= b // 8 < hi:
            m[b >> 5] |= 1 << (b & 31)
    st = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), 8, m)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value, device=dev)


torch.manual_seed(0)
stu = models.pose_resnet101(16, pretrained_backbone=False).to(dev)
tea = models.pose_resnet101(16, pretrained_backbone=False).to(dev)
stu.policy.update(pol); tea.policy.update(pol)
tr = MeanTeacherTrainer(stu, tea, precision="bf16")
b = synthetic.mean_teacher_batch(32, seed=0)
g = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in b.items()}
th = lambda ap: warp.recon_thetas(ap, 32, 4.0, dev)
ths, tht = th(g["aug_param_stu"]), th(g["aug_param_tea"])

kind, _, arg = cfg.partition(":")
nums = [int(v) for v in arg.split(",")] if arg else []
main = torch.cuda.current_stream()
if kind == "tea":
    tr._side = (masked_stream(0, nums[0]), torch.cuda.Stream(device=dev))
elif kind == "teax":
    tr._side = (masked_stream(0, nums[0]), masked_stream(nums[0], 32))
    main = masked_stream(nums[0], 32)
elif kind == "split":
    t, s = nums
    tr._side = (masked_stream(0, t), masked_stream(t, t + s))
    main = masked_stream(t + s, 32)
elif kind in ("wg", "wgx"):
    tr._wg_stream = [masked_stream(0, nums[0])]
    if kind == "wgx":
        tr._side = (torch.cuda.Stream(device=dev), masked_stream(nums[0], 32))
        main = masked_stream(nums[0], 32)
elif kind != "plain":
    raise SystemExit(__doc__)


def ev():
    e = torch.cuda.Event(enable_timing=True); e.record(); return e


orig_fw = stu.finish_wgrad
marks = {}
def fw(wg_stream=None):
    marks["chains_end"] = ev()
    return orig_fw(wg_stream)
stu.finish_wgrad = fw
rows = []
with torch.cuda.stream(main):
    for it in range(14):
        e0 = ev()
        st = tr._forward_part(g["x_s"], g["label_s"], g["weight_s"], g["x_t_stu"], [g["x_t_tea"]], ths, [tht])
        e1 = ev()
        tr._loss_backward_part(st, None)
        e2 = ev()
        tr._sync_grads(); tr._update()
        e3 = ev()
        torch.cuda.synchronize()
        rows.append((e0.elapsed_time(e1), e1.elapsed_time(marks["chains_end"]), marks["chains_end"].elapsed_time(e2), e2.elapsed_time(e3), e0.elapsed_time(e3)))
rows = rows[5:]
m = [sum(r[i] for r in rows) / len(rows) for i in range(5)]
print(f"{cfg:14s} {pol}: forwards {m[0]:.2f} ms | gradient chains {m[1]:.2f} | weight gradients after the chains {m[2]:.2f} | tail {m[3]:.2f} | "
      f"step {m[4]:.2f} (eager)", flush=True)
