"""Evaluation loop: mirror of mint/ctl/single_task_evaluator.py (row f1).  One eval step runs the
device-resident auto-regressive sampler for `steps` frames, prepends the seed motion and saves one
`{motion_name}_{audio_name}.npy` of shape (seed + generated, 225) per sample — the file contract
`tools/calculate_scores.py:210-215` consumes.

Under one-process-per-GPU data parallelism every rank takes its contiguous share of each global eval batch
(mint_amd/sharding.py; the reference's `strategy.run` does the same split, single_task_evaluator.py:86) and saves its own
files; no collective runs while sequences are generated."""
import os

import numpy as np
import torch

from mint_amd import sharding


class SingleTaskEvaluator:
    def __init__(self, eval_dataset, model, metrics=None, output_dir=None, evaluator_options=None, steps=1200,
                 rank=None, world_size=None):
        self.eval_dataset = eval_dataset
        self.model = model
        self.metrics = [] if metrics is None else (metrics if isinstance(metrics, list) else [metrics])
        self.output_dir = output_dir
        self.steps = steps
        # data-parallel evaluation: (rank, world) of the default process group unless given
        self.rank, self.world_size = (rank, world_size) if rank is not None and world_size is not None else (
            sharding.world_info())

    def eval_begin(self):
        for metric in self.metrics:
            metric.reset_states()

    def eval_step(self, iterator):
        """Returns (outputs of THIS rank's sequences, paths this rank saved).  Metrics, when there are any, are updated
        on every rank with the GLOBAL batch and the gathered global outputs, so eval_end() reports the same whole-set
        value everywhere (the reference's strategy-aware Keras metrics aggregate over replicas); FACT's own metric list
        is empty and the gather is skipped."""
        global_inputs = next(iterator)
        inputs = sharding.shard_inputs(global_inputs, self.rank, self.world_size)  # this replica's sequences
        total = int(global_inputs["motion_input"].shape[0])
        if int(inputs["motion_input"].shape[0]) == 0:  # more ranks than sequences in this batch
            if self.metrics and self.world_size > 1:  # still a party to the gather the other ranks run
                import torch.distributed as dist
                seed = torch.as_tensor(global_inputs["motion_input"])
                dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else seed.device
                empty = torch.zeros((0, int(seed.shape[1]) + self.steps, int(seed.shape[2])), dtype=seed.dtype, device=dev)
                full = sharding.gather_rows(empty, total, self.rank, self.world_size)
                for metric in self.metrics:
                    metric.update_state(global_inputs, full)
            return None, []
        # [batch, steps, dim] -> [batch, seed + steps, dim]   (single_task_evaluator.py:69-71)
        outputs = self.model.infer_auto_regressive(inputs, steps=self.steps)
        seed = torch.as_tensor(inputs["motion_input"]).to(outputs.device, outputs.dtype)
        outputs = torch.cat([seed, outputs], dim=1)
        paths = []
        if self.output_dir is not None:
            os.makedirs(self.output_dir, exist_ok=True)
            host = outputs.cpu().numpy()
            for i in range(host.shape[0]):
                path = os.path.join(self.output_dir, "%s_%s.npy" % (inputs["motion_name"][i], inputs["audio_name"][i]))
                np.save(path, host[i])
                paths.append(path)
        if self.metrics:
            if self.world_size > 1:
                full = sharding.gather_rows(outputs, total, self.rank, self.world_size)
                for metric in self.metrics:
                    metric.update_state(global_inputs, full)
            else:
                for metric in self.metrics:
                    metric.update_state(inputs, outputs)
        return outputs, paths

    def eval_end(self):
        return {metric.name: metric.result() for metric in self.metrics}

    def evaluate(self, num_steps=-1):
        """orbit.Controller.evaluate stand-in: one pass over the dataset."""
        self.eval_begin()
        it = iter(self.eval_dataset)
        n = 0
        while num_steps < 0 or n < num_steps:
            try:
                self.eval_step(it)
            except StopIteration:
                break
            n += 1
        return self.eval_end()
