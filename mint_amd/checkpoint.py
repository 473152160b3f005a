"""Checkpoint cadence of trainer.py:168-173 (tf.train.CheckpointManager: every `checkpoint_interval`
steps, keep `max_to_keep`) on the engine's state (fp32 params, Adam m/v, step) — row f3/f4.

The on-disk format is this repo's own (`ckpt-<step>.pt`: flat fp32 arenas + the variable-name table);
it is NOT a TensorFlow object-graph checkpoint.  `optimizer=None` mirrors the evaluator's
Checkpoint(model, global_step) (evaluator.py:64-67)."""
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
        if step is None:
            step = self.optimizer.iterations if self.optimizer is not None else int(self.model.global_step)
        if check_interval and step % self.checkpoint_interval != 0:
            return None
        state = self.model.state_dict()
        state["optimizer_iterations"] = int(self.optimizer.iterations if self.optimizer is not None
                                            else self.model.global_step)
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
        # the state holds only tensors, ints and lists of str: no pickled code is ever executed
        state = torch.load(path, map_location="cpu", weights_only=True)
        # like tf.train.Checkpoint the restore may precede variable creation (the reference restores
        # before the first call, trainer.py:168-173): the model applies it when the engine is built
        self.model.load_state_dict(state)
        if self.optimizer is not None:
            self.optimizer.iterations = int(state.get("optimizer_iterations", state.get("global_step", 0)))
        return path
