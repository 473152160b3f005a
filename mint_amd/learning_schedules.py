"""Learning-rate schedules of the reference (mint/core/learning_schedules.py) as plain Python
callables `schedule(step) -> float`; same class names, constructor arguments and error behaviour."""
import math


class ManualStepping:
    """Manual stepping learning rate schedule (learning_schedules.py:19-67)."""

    def __init__(self, lr_step_boundaries, lr_sequence, warmup, name=None):
        lr_step_boundaries = list(lr_step_boundaries)
        lr_sequence = list(lr_sequence)
        if any(b < 0 for b in lr_step_boundaries) or any(not isinstance(b, int) for b in lr_step_boundaries):
            raise ValueError("boundaries must be a list of positive integers")
        if any(bn <= b for bn, b in zip(lr_step_boundaries[1:], lr_step_boundaries[:-1])):
            raise ValueError("Entries in boundaries must be strictly increasing.")
        if any(not isinstance(r, float) for r in lr_sequence):
            raise ValueError("Learning rates must be floats")
        if len(lr_sequence) != len(lr_step_boundaries) + 1:
            raise ValueError("Number of provided learning rates must exceed "
                             "number of boundary points by exactly 1.")
        if lr_step_boundaries and lr_step_boundaries[0] == 0:
            raise ValueError("First step cannot be zero.")
        if warmup and lr_step_boundaries:
            slope = (lr_sequence[1] - lr_sequence[0]) * 1.0 / lr_step_boundaries[0]
            warmup_steps = list(range(lr_step_boundaries[0]))
            warmup_rates = [lr_sequence[0] + slope * step for step in warmup_steps]
            lr_step_boundaries = warmup_steps + lr_step_boundaries
            lr_sequence = warmup_rates + lr_sequence[1:]
        else:
            lr_step_boundaries = [0] + lr_step_boundaries
        self.num_boundaries = len(lr_step_boundaries)
        self.lr_step_boundaries = lr_step_boundaries
        self.lr_sequence = lr_sequence
        self.warmup = warmup
        self.name = name

    def __call__(self, step):
        # rate of the last boundary <= step (reduce_max over the where(), :61-67)
        step = int(step)
        rate_index = 0
        for i, b in enumerate(self.lr_step_boundaries):
            if step >= b:
                rate_index = max(rate_index, i)
        return float(self.lr_sequence[rate_index])


class PolynomialDecay:
    """tf.keras.optimizers.schedules.PolynomialDecay (used by trainer.py:64-70), cycle=False."""

    def __init__(self, initial_learning_rate, decay_steps, end_learning_rate=0.0001, power=1.0):
        self.initial_learning_rate = initial_learning_rate
        self.decay_steps = decay_steps
        self.end_learning_rate = end_learning_rate
        self.power = power

    def __call__(self, step):
        s = min(float(step), float(self.decay_steps))
        p = s / float(self.decay_steps)
        return ((self.initial_learning_rate - self.end_learning_rate) * (1.0 - p) ** self.power +
                self.end_learning_rate)


class WarmUp:
    """Polynomial warmup wrapped around a decay schedule (learning_schedules.py:70-125)."""

    def __init__(self, initial_learning_rate, decay_schedule_fn, warmup_steps, power=1.0, name=None):
        self.initial_learning_rate = initial_learning_rate
        self.warmup_steps = warmup_steps
        self.power = power
        self.decay_schedule_fn = decay_schedule_fn
        self.name = name

    def __call__(self, step):
        g, w = float(step), float(self.warmup_steps)
        if g < w:
            return self.initial_learning_rate * (g / w) ** self.power
        return self.decay_schedule_fn(step - self.warmup_steps)

    def get_config(self):
        return {"initial_learning_rate": self.initial_learning_rate,
                "decay_schedule_fn": self.decay_schedule_fn, "warmup_steps": self.warmup_steps,
                "power": self.power, "name": self.name}


class CosineDecayWithWarmup:
    """Keras CosineDecay with a linear warmup (learning_schedules.py:128-175).

    After warmup it evaluates Keras' CosineDecay at s = step - warmup + 1 with
    decay_steps = steps - warmup: lr0 * ((1-alpha) * 0.5*(1+cos(pi*min(s, D)/D)) + alpha).
    (As checked in, the reference class derives from LearningRateSchedule and its super().__call__
    is abstract; decay_steps = steps - warmup is what reproduces the reference's own golden vector
    in learning_schedules_test.py:22-40, which tests/test_learning_schedules.py pins.)"""

    def __init__(self, initial_learning_rate, steps, warmup=0, alpha=0.0):
        self.initial_learning_rate = initial_learning_rate
        self.steps = steps
        self.warmup = warmup
        self.alpha = alpha
        self.name = None

    def __call__(self, step):
        g, w = float(step), float(self.warmup)
        if g < w:
            return g * self.initial_learning_rate / (w - 1.0)
        decay_steps = float(self.steps) - w
        s = min(g - w + 1.0, decay_steps)
        cosine = 0.5 * (1.0 + math.cos(math.pi * s / decay_steps))
        return self.initial_learning_rate * ((1.0 - self.alpha) * cosine + self.alpha)

    def get_config(self):
        return {"initial_learning_rate": self.initial_learning_rate, "steps": self.steps,
                "warmup": self.warmup, "alpha": self.alpha, "name": self.name}


def create_learning_rate(learning_rate_config, initial_learning_rate=0.1, warmup_steps=1000):
    """trainer.py:49-96 `_create_learning_rate` (flags become keyword arguments)."""
    lr_schedule = None
    kind = learning_rate_config.WhichOneof("learning_rate")
    if kind == "exponential_decay_learning_rate":
        config = learning_rate_config.exponential_decay_learning_rate
        lr_schedule = PolynomialDecay(initial_learning_rate, decay_steps=config.decay_steps,
                                      end_learning_rate=config.min_learning_rate, power=config.decay_factor)
        if warmup_steps:
            lr_schedule = WarmUp(initial_learning_rate, decay_schedule_fn=lr_schedule,
                                 warmup_steps=warmup_steps)
    if kind == "manual_step_learning_rate":
        config = learning_rate_config.manual_step_learning_rate
        if not config.schedule:
            raise ValueError("Empty learning rate schedule.")
        boundaries = [x.step for x in config.schedule]
        sequence = [config.initial_learning_rate] + [x.learning_rate for x in config.schedule]
        lr_schedule = ManualStepping(boundaries, sequence, config.warmup)
    if kind == "cosine_decay_learning_rate":
        config = learning_rate_config.cosine_decay_learning_rate
        lr_schedule = CosineDecayWithWarmup(initial_learning_rate, config.total_steps, warmup_steps)
    if lr_schedule is None:
        raise ValueError("Learning_rate %s not supported." % kind)
    return lr_schedule
