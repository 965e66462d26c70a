#!/usr/bin/env python3
"""CLI of the reference's pointNet/self-attention/test_pointnet_att_segmen.py (:287-305, same flags) on the HIP path,
plus --cluster_dir (the reference hard-codes `k_means_25/`, :140-143)."""
import argparse
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
test = importlib.import_module("3d-semantic-segmentation-amp-net_amd.pointNet.amp_test").test

if __name__ == '__main__':
    parser = argparse.ArgumentParser()
    parser.add_argument('--dataset_path', type=str, default='/dades/LIDAR/towers_detection/datasets/towers_100x100')
    parser.add_argument('--out_path', type=str, default='results')
    parser.add_argument('--number_of_points', type=int, default=2048)
    parser.add_argument('--number_of_workers', type=int, default=0)
    parser.add_argument('--model_checkpoint', type=str, default='')
    parser.add_argument('--path_list_files', type=str, default='train_test_files/RGBN_100x100')
    parser.add_argument('--cluster_dir', type=str, default='k_means_25')
    a = parser.parse_args()
    test(a.dataset_path, a.out_path, a.number_of_points, a.number_of_workers, a.model_checkpoint, a.path_list_files, a.cluster_dir)
