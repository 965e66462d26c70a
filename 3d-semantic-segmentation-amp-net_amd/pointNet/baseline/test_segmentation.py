"""CLI of the reference's pointNet/baseline/test_segmentation.py, same flags; the work is baseline_seg.test."""
import argparse
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))

if __name__ == '__main__':
    p = argparse.ArgumentParser()
    p.add_argument('dataset_folder', type=str)
    p.add_argument('--output_folder', type=str, default='pointNet/results')
    p.add_argument('--number_of_points', type=int, default=2048)
    p.add_argument('--number_of_workers', type=int, default=0)
    p.add_argument('--model_checkpoint', type=str, required=True)
    p.add_argument('--path_list_files', type=str, default='pointNet/data/train_test_files/RGBN')
    p.add_argument('--model', choices=['pointnet', 'light'], default='pointnet')
    a = p.parse_args()
    B = importlib.import_module("3d-semantic-segmentation-amp-net_amd.pointNet.baseline_seg")
    B.test(a.dataset_folder, a.number_of_points, a.output_folder, a.number_of_workers, a.model_checkpoint, a.path_list_files, a.model)
