"""Checkpoint cadence of trainer.py:168-173 (tf.train.CheckpointManager: every `checkpoint_interval`
steps, keep `max_to_keep`) on the engine's state (fp32 params, Adam m/v, step) — row f3/f4."""
import glob
import os
import re

import torch


class CheckpointManager:
    def __init__(self, model, optimizer, directory, checkpoint_interval=1000, max_to_keep=5):
        self.model, self.optimizer, self.directory = model, optimizer, directory
        self.checkpoint_interval, self.max_to_keep = checkpoint_interval, max_to_keep
        os.makedirs(directory, exist_ok=True)

    def _paths(self):
        found = []
        for p in glob.glob(os.path.join(self.directory, "ckpt-*.pt")):
            m = re.search(r"ckpt-(\d+)\.pt$", p)
            if m:
                found.append((int(m.group(1)), p))
        return sorted(found)

    @property
    def latest_checkpoint(self):
        paths = self._paths()
        return paths[-1][1] if paths else None

    def save(self, step=None, check_interval=True):
        step = self.optimizer.iterations if step is None else step
        if check_interval and step % self.checkpoint_interval != 0:
            return None
        state = self.model.state_dict()
        state["optimizer_iterations"] = int(self.optimizer.iterations)
        path = os.path.join(self.directory, "ckpt-%d.pt" % step)
        torch.save(state, path + ".tmp")
        os.replace(path + ".tmp", path)
        for _, old in self._paths()[:-self.max_to_keep]:
            os.remove(old)
        return path

    def restore_or_initialize(self):
        path = self.latest_checkpoint
        if path is None:
            return None
        state = torch.load(path, map_location="cpu", weights_only=False)
        self.model.load_state_dict(state)
        self.optimizer.iterations = int(state.get("optimizer_iterations", state.get("global_step", 0)))
        return path
