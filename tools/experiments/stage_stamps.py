"""Dev tool: where one EAGER mean-teacher step spends its time on the device (HIP events on the main stream): forwards | both gradient
chains | what is left of the weight gradients after the chains have ended (the exposed tail) | optimizer tail - for the unstaged step and
for staged / residency-capped weight gradients (policy wgrad_overlap / wgrad_cap).  usage: python tools/stage_stamps.py [FIELD=INT ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uda_poseestimation_amd import synthetic, warp
from uda_poseestimation_amd.engine import MeanTeacherTrainer
import uda_poseestimation_amd.lib.models as models

pol = {k: int(v) for k, v in (a.split("=") for a in sys.argv[1:] if "=" in a and not a.startswith("--"))}
nstreams = int(os.environ.get("WGS", "1"))
dev = torch.device("cuda:0")
torch.manual_seed(0)
stu = models.pose_resnet101(16, pretrained_backbone=False).to(dev)
tea = models.pose_resnet101(16, pretrained_backbone=False).to(dev)
stu.policy.update(pol); tea.policy.update(pol)
tr = MeanTeacherTrainer(stu, tea, precision="bf16")
tr.wgrad_streams = nstreams
b = synthetic.mean_teacher_batch(32, seed=0)
g = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in b.items()}
th = lambda ap: warp.recon_thetas(ap, 32, 4.0, dev)
ths, tht = th(g["aug_param_stu"]), th(g["aug_param_tea"])

def ev():
    e = torch.cuda.Event(enable_timing=True); e.record(); return e

import types
orig_fw = stu.finish_wgrad
marks = {}
def fw(wg_stream=None):
    marks["chains_end"] = ev()          # (main has waited for both chains here)
    return orig_fw(wg_stream)
stu.finish_wgrad = fw
rows = []
for it in range(12):
    e0 = ev()
    st = tr._forward_part(g["x_s"], g["label_s"], g["weight_s"], g["x_t_stu"], [g["x_t_tea"]], ths, [tht])
    e1 = ev()
    tr._loss_backward_part(st, None)
    e2 = ev()
    tr._sync_grads(); tr._update()
    e3 = ev()
    torch.cuda.synchronize()
    rows.append((e0.elapsed_time(e1), e1.elapsed_time(marks["chains_end"]), marks["chains_end"].elapsed_time(e2), e2.elapsed_time(e3), e0.elapsed_time(e3)))
rows = rows[4:]
m = [sum(r[i] for r in rows) / len(rows) for i in range(5)]
print(f"policy {pol} streams {nstreams}: forwards {m[0]:.2f} ms | gradient chains {m[1]:.2f} | weight gradients after the chains {m[2]:.2f} | tail {m[3]:.2f} | step {m[4]:.2f} (eager, one process)")
