"""Checkpoint cadence of trainer.py:168-173 (tf.train.CheckpointManager: every `checkpoint_interval`
steps, keep `max_to_keep`) on the engine's state (fp32 params, Adam m/v, step) — row f3/f4.

The manager's on-disk format is this repo's own (`ckpt-<step>.pt`: flat fp32 arenas + the variable-name table).
The reference's own format - TensorFlow object-graph checkpoints - is read and written by
`import_tf_checkpoint` / `export_tf_checkpoint` below (format code in mint_amd/tf_checkpoint.py), so weights trained
with the reference can be evaluated here and vice versa.  `optimizer=None` mirrors the evaluator's
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


# ---- TensorFlow object-graph checkpoints (the reference's own format, trainer.py:168-173 / evaluator.py:64-67) --------
def import_tf_checkpoint(model, prefix_or_dir, optimizer=None, verify_crc=False):
    """Load a checkpoint written by the reference (`tf.train.Checkpoint(optimizer=, model=)` / CheckpointManager) into a
    BUILT mint_amd FACTModel: weights, and - when the checkpoint holds them and the model trains - the Adam slots and
    the iteration counter.  `prefix_or_dir`: a checkpoint prefix (`.../ckpt-1000`) or a directory with a `checkpoint`
    state file.  Returns the prefix read.  Format notes and verification status: mint_amd/tf_checkpoint.py."""
    import os
    import numpy as np
    from mint_amd import tf_checkpoint as T
    prefix = T.latest_checkpoint(prefix_or_dir) if os.path.isdir(prefix_or_dir) else prefix_or_dir
    if prefix is None:
        raise FileNotFoundError("no `checkpoint` state file in %s" % prefix_or_dir)
    names = model.variable_names
    views = model.trainable_variables
    ck = T.read_fact_checkpoint(prefix, names, {n: tuple(v.shape) for n, v in zip(names, views)}, verify_crc)
    for n, v in zip(names, views):
        v.copy_(torch.from_numpy(np.ascontiguousarray(ck["params"][n])).view(v.shape))
    if ck["adam_m"] is not None and model.is_training:
        for arena, slot in (("adam_m", ck["adam_m"]), ("adam_v", ck["adam_v"])):
            for n, v in zip(names, model._views(arena)):
                v.copy_(torch.from_numpy(np.ascontiguousarray(slot[n])).view(v.shape))
    step = ck["iterations"] if ck["iterations"] is not None else ck["global_step"]
    if step is not None:
        model.global_step = int(step)
        if optimizer is not None:
            optimizer.iterations = int(step)
        from mint_amd import _lib as L
        L.check(L.lib().fact_set_step(model._h, int(step)))
    model.sync_weights()
    return prefix


def export_tf_checkpoint(model, prefix, optimizer=None):
    """Write the model (and, for a training model, Adam m / v + the iteration counter) as an object-graph checkpoint
    with the reference's variable paths, restorable by its trainer / evaluator."""
    from mint_amd import tf_checkpoint as T
    names = model.variable_names
    get = lambda arena: {n: v.detach().cpu().numpy() for n, v in zip(names, model._views(arena))}
    train = model.is_training
    it = optimizer.iterations if optimizer is not None else int(model.global_step)
    T.write_fact_checkpoint(prefix, get("params"), get("adam_m") if train else None, get("adam_v") if train else None,
                            iterations=it if train else None)
    return prefix
