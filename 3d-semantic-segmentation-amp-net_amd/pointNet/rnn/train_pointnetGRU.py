#!/usr/bin/env python3
"""CLI of the reference's pointNet/rnn/train_pointnetGRU.py (:444-483, same flags and defaults) on the HIP path.
Run from the repository root:  python <package>/pointNet/rnn/train_pointnetGRU.py --dataset_path DATASET ..."""
import argparse
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
train_gru = importlib.import_module("3d-semantic-segmentation-amp-net_amd.pointNet.gru_train").train_gru

if __name__ == '__main__':
    parser = argparse.ArgumentParser()
    parser.add_argument('--task', type=str, choices=['classification', 'segmentation'], default='segmentation')
    parser.add_argument('--dataset_path', type=str, default='/dades/LIDAR/towers_detection/datasets/kmeans_100x100c9_2048')
    parser.add_argument('--path_list_files', type=str, default='train_test_files/RGBN_100x100_old/RGBN_100x100_kmeans')
    parser.add_argument('--output_folder', type=str, default='pointNet/results')
    parser.add_argument('--number_of_points', type=int, default=2048)
    parser.add_argument('--number_of_windows', type=int, default=9)
    parser.add_argument('--batch_size', type=int, default=32)
    parser.add_argument('--epochs', type=int, default=500)
    parser.add_argument('--learning_rate', type=float, default=0.0005)
    parser.add_argument('--weighing_method', type=str, default='EFS')
    parser.add_argument('--beta', type=float, default=0.999)
    parser.add_argument('--number_of_workers', type=int, default=0)
    parser.add_argument('--model_checkpoint', type=str, default='')
    parser.add_argument('--c_sample', type=bool, default=False)
    a = parser.parse_args()
    train_gru(a.task, a.dataset_path, a.path_list_files, a.output_folder, a.number_of_points, a.number_of_windows, a.batch_size, a.epochs,
              a.learning_rate, a.weighing_method, a.beta, a.number_of_workers, a.model_checkpoint, a.c_sample)
