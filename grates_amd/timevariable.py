"""
Time-variable gravity fields as sums of constituents (drop-in names of grates/gravityfield.py:784-812, 1054-1140: `Trend`,
`Oscillation`, `TimeVariableGravityField`).  Host-side containers: a constituent is anything with `evaluate_at(epoch)`; the device
work happens where the evaluated fields are used (`gravityfield.gridded_rms`, batched synthesis).

Every constituent here is a weighted sum of fixed fields, V(t) = sum_i w_i(tau) V_i with tau the time since the reference epoch
in the constituent's own unit -- `Trend` has one term with w = tau, `Oscillation` two with w = (cos, sin)(2 pi tau).
"""

import functools
import math


class _WeightedFields:
    """V(t) = sum_i w_i(tau(t)) V_i; the fields are copied on construction (anything with `copy`, `*` by a float and `+`)."""

    def __init__(self, fields, reference_epoch, unit_days):
        self._fields = [f.copy() for f in fields]
        self._t0 = reference_epoch
        self._unit = 86400.0 * unit_days

    def _weights(self, tau):
        raise NotImplementedError

    def evaluate_at(self, epoch):
        tau = (epoch - self._t0).total_seconds() / self._unit
        terms = [field * weight for field, weight in zip(self._fields, self._weights(tau))]
        out = functools.reduce(lambda acc, term: acc + term, terms[1:], terms[0])
        out.epoch = epoch
        return out


class Trend(_WeightedFields):
    """Linear trend V(t) = V (t - t0) / time_scale, `time_scale` in days (default: a Julian year), grates/gravityfield.py:1054-1094."""

    def __init__(self, gravity_field, reference_epoch, time_scale=365.25):
        super().__init__([gravity_field], reference_epoch, time_scale)

    def _weights(self, tau):
        return (tau,)


class Oscillation(_WeightedFields):
    """V(t) = V_c cos(2 pi (t - t0) / T) + V_s sin(2 pi (t - t0) / T), period T in days, grates/gravityfield.py:1097-1140."""

    def __init__(self, gravity_field_cosine, gravity_field_sine, period, reference_epoch):
        super().__init__([gravity_field_cosine, gravity_field_sine], reference_epoch, period)

    def _weights(self, tau):
        phase = 2.0 * math.pi * tau
        return (math.cos(phase), math.sin(phase))


class TimeVariableGravityField:
    """Sum of constituents (trend, cycles, an interpolated `TimeSeries`, ...), each with `evaluate_at`; `constituents` stays a
    public list like upstream (grates/gravityfield.py:784-812)."""

    def __init__(self, constituents):
        self.constituents = constituents

    def evaluate_at(self, epoch):
        parts = [c.evaluate_at(epoch) for c in self.constituents]
        return functools.reduce(lambda acc, part: acc + part, parts[1:], parts[0])
