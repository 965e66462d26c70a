#!/usr/bin/env python3
"""CLI of the reference's pointNet/rnn/test_pointnet_gru_segmen.py (:253-284, same flags and defaults) on the HIP path."""
import argparse
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
test = importlib.import_module("3d-semantic-segmentation-amp-net_amd.pointNet.gru_train").test

if __name__ == '__main__':
    parser = argparse.ArgumentParser()
    parser.add_argument('--dataset_path', type=str, default='/dades/LIDAR/towers_detection/datasets/towers_100x100')
    parser.add_argument('--out_path', type=str, default='results')
    parser.add_argument('--number_of_points', type=int, default=1024)
    parser.add_argument('--number_of_workers', type=int, default=0)
    parser.add_argument('--model_checkpoint', type=str, default='')
    parser.add_argument('--path_list_files', type=str, default='train_test_files/RGBN_100x100_old')
    a = parser.parse_args()
    test(a.dataset_path, a.out_path, a.number_of_points, a.number_of_workers, a.model_checkpoint, a.path_list_files)
