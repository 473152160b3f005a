"""Learning-rate schedules against the reference's own golden vector
(mint/core/learning_schedules_test.py:22-40) and its constructor error behaviour
(mint/core/learning_schedules.py:24-43)."""
import pytest

from mint_amd import learning_schedules as ls
from mint_amd import protos


@pytest.mark.parametrize("cast", [int, float])
def test_cosine_with_warmup_reference_kat(cast):
    lr = ls.CosineDecayWithWarmup(initial_learning_rate=1.0, steps=10, warmup=4, alpha=1e-4)
    got = [lr(cast(i)) for i in range(10)]
    exp = [0.0, 0.33, 0.66, 1.0, 0.933, 0.750, 0.500, 0.25, 0.067, 1e-04]
    for g, e in zip(got, exp):
        assert abs(g - e) <= 1e-2 + 1e-2 * abs(e)


def test_manual_stepping_fact_v5_schedule():
    s = ls.ManualStepping([100000, 150000], [1e-4, 1e-5, 1e-6], False)
    assert [s(x) for x in (0, 1, 99999, 100000, 149999, 150000, 2400000)] == [1e-4, 1e-4, 1e-4, 1e-5, 1e-5, 1e-6, 1e-6]


def test_manual_stepping_warmup_interpolates():
    s = ls.ManualStepping([10], [0.0, 1.0], True)
    assert s(0) == 0.0 and abs(s(5) - 0.5) < 1e-12 and s(10) == 1.0 and s(1000) == 1.0


@pytest.mark.parametrize("args", [([-1], [1.0, 2.0]), ([5, 5], [1.0, 2.0, 3.0]), ([5], [1, 2.0]), ([5], [1.0]),
                                  ([0, 5], [1.0, 2.0, 3.0])])
def test_manual_stepping_errors(args):
    with pytest.raises(ValueError):
        ls.ManualStepping(args[0], args[1], False)


def test_warmup_and_polynomial():
    base = ls.PolynomialDecay(1.0, decay_steps=100, end_learning_rate=0.0, power=1.0)
    w = ls.WarmUp(1.0, base, warmup_steps=10)
    assert w(0) == 0.0 and abs(w(5) - 0.5) < 1e-12 and abs(w(10) - 1.0) < 1e-12 and abs(w(60) - 0.5) < 1e-12


def test_create_learning_rate_from_config():
    lr = protos.LearningRate()
    m = lr.manual_step_learning_rate
    m.initial_learning_rate = 1e-4
    m.schedule.add(step=100000, learning_rate=1e-5)
    sched = ls.create_learning_rate(lr)
    assert sched(0) == 1e-4 and sched(100000) == 1e-5
    empty = protos.LearningRate()
    empty.manual_step_learning_rate.initial_learning_rate = 1.0
    with pytest.raises(ValueError):
        ls.create_learning_rate(empty)
    with pytest.raises(ValueError):
        ls.create_learning_rate(protos.LearningRate())
