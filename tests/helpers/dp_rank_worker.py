"""Test infrastructure (tests/test_gpu_hotpath.py::test_two_rank_synced_gradient_is_the_mean_of_the_rank_gradients): ONE data-parallel
mean-teacher forward / backward / gradient exchange on each of two ranks that share cuda:0 and talk over gloo; every rank writes its
synchronised flat gradient, its confidences and its consistency mask to <outdir>/rank<r>.npz.  Launched with torch.distributed.run."""
import os
import sys

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def build(seed=0):
    import uda_poseestimation_amd.lib.models.pose_resnet as pr
    pr.PoseResNet.default_precision = "bf16"
    torch.manual_seed(seed)
    stu = pr._pose_resnet("t", 16, pr.Bottleneck_default, [1, 2, 2, 1], False, False).cuda()
    torch.manual_seed(seed)
    tea = pr._pose_resnet("t", 16, pr.Bottleneck_default, [1, 2, 2, 1], False, False).cuda()
    return stu, tea


def shard(rank, N=4, S=128):
    from uda_poseestimation_amd import synthetic
    b = synthetic.mean_teacher_batch(N, num_keypoints=16, image_size=S, heatmap_size=S // 4, seed=10 + rank)
    return {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in b.items()}


def main():
    out = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from uda_poseestimation_amd import warp
    from uda_poseestimation_amd.engine import MeanTeacherTrainer
    stu, tea = build()
    tr = MeanTeacherTrainer(stu, tea, image_size=128, heatmap_size=32)
    g = shard(rank)
    th = lambda ap: warp.recon_thetas(ap, 4, 4.0, "cuda")
    res = tr._forward_backward(g["x_s"], g["label_s"], g["weight_s"], g["x_t_stu"], [g["x_t_tea"]], th(g["aug_param_stu"]), [th(g["aug_param_tea"])])
    tr._sync_grads()
    torch.cuda.synchronize()
    flat = stu._flat_grad.detach().cpu().numpy()
    np.savez(os.path.join(out, f"rank{rank}.npz"), flat=flat, mask=res["tea_mask"].cpu().numpy(), loss=float(res["loss_all"]))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
