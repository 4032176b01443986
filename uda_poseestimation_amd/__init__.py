"""MI355X-native hot path of VisionLearningGroup/UDA_PoseEstimation (mean-teacher UDA pose estimation).

Drop-in module layout mirroring the reference (put this directory on sys.path to `import lib.models` unchanged):
    lib/models/{pose_resnet,resnet,loss,Style_net}.py, lib/keypoint_detection.py, utils.py
All device work runs in hand-written gfx950 HIP kernels reached through the C ABI of libudapose_hip.so.
"""
__version__ = "0.1.0"
