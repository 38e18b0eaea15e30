"""Boxes / patches chosen for every unit of the BASELINE trunk (MD_PLAN_PRINT=1 makes the library print them at plan time):
    MD_PLAN_PRINT=1 python tools/plan_print.py"""
import os, sys
os.environ["MD_PLAN_PRINT"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'disruption-prediciton-based-on-multimodal-deep-learning_amd'))
import torch
from src.models.R2Plus1D import R2Plus1DClassifier
from src.loss import FocalLoss
m = R2Plus1DClassifier(input_size=(3, 21, 128, 128), num_classes=2, layer_sizes=[1, 2, 2, 1], alpha=0.01).cuda().train()
x = torch.randn(8, 3, 21, 128, 128, device="cuda"); y = torch.zeros(8, dtype=torch.long, device="cuda")
FocalLoss(weight=torch.ones(2), gamma=2.0)(m(x), y).backward()
torch.cuda.synchronize()
