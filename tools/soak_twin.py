"""Dev tool (round 5): repeated twin runs of the captured mean-teacher step - same seeds, same batches, two independently built trainers - compared
step by step.  Two runs may differ by the arrival order of fp32 atomics (stem / split weight gradients, re-warp backward): ~1e-7 relative per step;
anything larger in the first steps is a race or an ordering fault (the memset-node fault of profiles/r5_ab_runs.txt 9 showed up here as
percent-level differences in a few percent of the runs).  usage: python tools/soak_twin.py [reps] [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from uda_poseestimation_amd import synthetic
from uda_poseestimation_amd.engine import GraphedTrainStep, MeanTeacherTrainer
import uda_poseestimation_amd.lib.models.pose_resnet as pr

pr.PoseResNet.default_precision = "bf16"
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
N, K, S = 4, 16, 128


def tiny(seed):
    torch.manual_seed(seed)
    return pr._pose_resnet("t", K, pr.Bottleneck_default, [1, 2, 2, 1], False, False)


def run(seed, occl):
    stu, tea = tiny(seed).cuda(), tiny(seed).cuda()
    tr = MeanTeacherTrainer(stu, tea, lr=1e-4, image_size=S, heatmap_size=S // 4, rng=np.random.RandomState(seed),
                            **({"occlude_rate": 0.5, "occlude_thresh": 0.0} if occl else {}))
    tr.device_occlusion = bool(occl)
    b = synthetic.mean_teacher_batch(N, num_keypoints=K, image_size=S, heatmap_size=S // 4, seed=seed)
    g = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in b.items()}
    args = (g["x_s"], g["label_s"], g["weight_s"], g["x_t_stu"], g["x_t_tea"], g["aug_param_stu"], g["aug_param_tea"])
    gs = GraphedTrainStep(tr, *args, warmup=1)
    out = []
    for it in range(steps):
        o = gs.step(*args)
        _ = (o["loss_all"] * 2).abs().max().item()          # (eager device work between replays, as a training loop has)
        out.append(float(o["loss_all"]))
    return out


worst = 0.0
for rep in range(reps):
    occl = bool(rep & 1)
    a, b = run(100 + rep, occl), run(100 + rep, occl)
    rel = [abs(x - y) / max(abs(x), 1e-12) for x, y in zip(a, b)]
    worst = max(worst, max(rel[:4]))
    print(f"rep {rep:2d} occlusion {int(occl)}: loss {a[0]:.6f} .. {a[-1]:.6f} | relative twin difference per step: " + " ".join(f"{r:.1e}" for r in rel), flush=True)
print(f"worst relative difference over the first four steps: {worst:.2e}")
