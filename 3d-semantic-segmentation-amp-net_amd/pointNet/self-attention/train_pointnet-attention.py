#!/usr/bin/env python3
"""CLI of the reference's pointNet/self-attention/train_pointnet-attention.py (:478-512, same flags and defaults)
on the HIP path.  Run from the repository root:  python <package>/pointNet/self-attention/train_pointnet-attention.py DATASET ...
Multi-GPU: python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 <this file> DATASET ..."""
import argparse
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
train_att = importlib.import_module("3d-semantic-segmentation-amp-net_amd.pointNet.amp_train").train_att

if __name__ == '__main__':
    parser = argparse.ArgumentParser()
    parser.add_argument('dataset_path', type=str, help='path to the dataset folder')
    parser.add_argument('--task', type=str, choices=['classification', 'segmentation'], default='segmentation')
    parser.add_argument('--path_list_files', type=str, default='train_test_files/RGBN_100x100')
    parser.add_argument('--out_path', type=str, default='pointNet/results')
    parser.add_argument('--number_of_points', type=int, default=2048)
    parser.add_argument('--number_of_windows', type=int, default=9)
    parser.add_argument('--batch_size', type=int, default=32)
    parser.add_argument('--epochs', type=int, default=500)
    parser.add_argument('--learning_rate', type=float, default=0.001)
    parser.add_argument('--weighing_method', type=str, default='EFS')
    parser.add_argument('--beta', type=float, default=0.999)
    parser.add_argument('--number_of_workers', type=int, default=8)
    parser.add_argument('--model_checkpoint', type=str, default='')
    a = parser.parse_args()
    train_att(a.task, a.dataset_path, a.path_list_files, a.out_path, a.number_of_points, a.batch_size, a.epochs, a.learning_rate,
              a.weighing_method, a.beta, a.number_of_workers, a.model_checkpoint)
