"""Checkpoint I/O in the reference's on-disk layout (reference code/training/idr_train.py:30-38, 130-144, 164-184):

    <checkpoints_path>/ModelParameters/{<epoch>,latest}.pth       {"epoch", "model_state_dict"}
    <checkpoints_path>/OptimizerParameters/{<epoch>,latest}.pth   {"epoch", "optimizer_state_dict"}
    <checkpoints_path>/SchedulerParameters/{<epoch>,latest}.pth   {"epoch", "scheduler_state_dict"}

The model's state_dict keys are the reference's (implicit_network.lin{l}.{bias,weight_g,weight_v}, rendering_network.lin{l}.*) and
optim.FlatAdam speaks torch.optim.Adam's state layout, so files written by the reference load here and vice versa (camera-training files
-- OptimizerCamParameters / CamParameters -- belong to the reference's disabled train_cameras branch and are not handled)."""
import os

import torch

MODEL_SUBDIR, OPTIMIZER_SUBDIR, SCHEDULER_SUBDIR = 'ModelParameters', 'OptimizerParameters', 'SchedulerParameters'


def _save_both(obj, directory, epoch):
    os.makedirs(directory, exist_ok=True)
    torch.save(obj, os.path.join(directory, str(epoch) + '.pth'))
    torch.save(obj, os.path.join(directory, 'latest.pth'))


def save_checkpoints(checkpoints_path, epoch, model, optimizer=None, scheduler=None):
    """idr_train.py:164-184: every file twice, as <epoch>.pth and latest.pth."""
    _save_both({'epoch': epoch, 'model_state_dict': model.state_dict()}, os.path.join(checkpoints_path, MODEL_SUBDIR), epoch)
    if optimizer is not None:
        _save_both({'epoch': epoch, 'optimizer_state_dict': optimizer.state_dict()}, os.path.join(checkpoints_path, OPTIMIZER_SUBDIR), epoch)
    if scheduler is not None:
        _save_both({'epoch': epoch, 'scheduler_state_dict': scheduler.state_dict()}, os.path.join(checkpoints_path, SCHEDULER_SUBDIR), epoch)


def load_checkpoints(checkpoints_path, model, optimizer=None, scheduler=None, checkpoint='latest', map_location=None):
    """idr_train.py:130-144 (--is_continue): -> start epoch.  `checkpoint`: 'latest' or an epoch number."""
    name = str(checkpoint) + '.pth'
    saved = torch.load(os.path.join(checkpoints_path, MODEL_SUBDIR, name), map_location=map_location)
    model.load_state_dict(saved['model_state_dict'])
    if optimizer is not None:
        data = torch.load(os.path.join(checkpoints_path, OPTIMIZER_SUBDIR, name), map_location=map_location)
        optimizer.load_state_dict(data['optimizer_state_dict'])
    if scheduler is not None:
        data = torch.load(os.path.join(checkpoints_path, SCHEDULER_SUBDIR, name), map_location=map_location)
        scheduler.load_state_dict(data['scheduler_state_dict'])
    return saved['epoch']
