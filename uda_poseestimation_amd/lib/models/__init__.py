from .pose_resnet import *
from . import loss

__all__ = ['pose_resnet']
