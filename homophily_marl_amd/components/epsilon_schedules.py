"""Epsilon schedule (reference: src/components/epsilon_schedules.py:4-25)."""
import math


class DecayThenFlatSchedule:
    def __init__(self, start, finish, time_length, decay="exp"):
        self.start, self.finish, self.time_length, self.decay = start, finish, time_length, decay
        self.delta = (start - finish) / time_length
        if decay == "exp":
            self.exp_scaling = -time_length / math.log(finish) if finish > 0 else 1

    def eval(self, T):
        if self.decay == "linear":
            return max(self.finish, self.start - self.delta * T)
        if self.decay == "exp":
            return min(self.start, max(self.finish, math.exp(-T / self.exp_scaling)))
        raise ValueError(self.decay)
