"""src/temperature_ramps.jl: callable ramps used as NVT.ktemp(step) with 1-based steps."""
import math


class LinearRamp:
    """src/temperature_ramps.jl:7-29"""

    def __init__(self, T_initial, T_final, n_steps):
        self.T_initial, self.T_final, self.n_steps = float(T_initial), float(T_final), int(n_steps)

    def __call__(self, step):
        if step > self.n_steps:
            return self.T_final
        step = min(max(step, 1), self.n_steps)
        if self.n_steps == 1:
            return self.T_final
        progress = (step - 1) / (self.n_steps - 1)
        return self.T_initial + (self.T_final - self.T_initial) * progress


class ExponentialRamp:
    """src/temperature_ramps.jl:36-60"""

    def __init__(self, T_initial, T_final, n_steps):
        self.T_initial, self.T_final, self.n_steps = float(T_initial), float(T_final), int(n_steps)

    def __call__(self, step):
        if step > self.n_steps:
            return self.T_final
        step = min(max(step, 1), self.n_steps)
        if self.n_steps == 1 or self.T_initial == self.T_final:
            return self.T_final
        progress = (step - 1) / (self.n_steps - 1)
        alpha = math.log(self.T_final / self.T_initial)
        return self.T_initial * math.exp(alpha * progress)


def initial_temperature_for_velocities(ktemp):
    """src/temperature_ramps.jl:67-73"""
    if hasattr(ktemp, "T_initial") and hasattr(ktemp, "T_final"):
        return max(ktemp.T_initial, ktemp.T_final)
    return ktemp
