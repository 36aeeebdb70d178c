"""Action selection (reference: src/components/action_selectors.py:35-68, EpsilonGreedyActionSelector).

epsilon-greedy over masked Q: with probability epsilon a uniformly random AVAILABLE action, else argmax of the masked
Q-values.  The torch RNG stream is not parity-pinned between CPU and GPU: parity tests feed actions."""
import torch as th

from .epsilon_schedules import DecayThenFlatSchedule

REGISTRY = {}


class EpsilonGreedyActionSelector:
    def __init__(self, args):
        self.args = args
        self.schedule = DecayThenFlatSchedule(args.epsilon_start, args.epsilon_finish, args.epsilon_anneal_time, decay="linear")
        self.epsilon = self.schedule.eval(0)

    def select_action(self, agent_inputs, avail_actions, t_env, test_mode=False):
        self.epsilon = self.schedule.eval(t_env)
        zero_after = getattr(self.args, "epsilon_zero", None)
        if zero_after is not None and t_env > zero_after:
            self.epsilon = 0.0
        if test_mode:
            self.epsilon = 0.0
        q = agent_inputs.masked_fill(avail_actions == 0, -float("inf"))
        greedy = q.argmax(dim=-1)
        if self.epsilon <= 0.0:
            return greedy
        explore = th.rand_like(agent_inputs.select(-1, 0)) < self.epsilon
        flat = avail_actions.reshape(-1, avail_actions.shape[-1]).float()
        rand = th.multinomial(flat, 1, replacement=True).reshape(avail_actions.shape[:-1])
        return th.where(explore, rand, greedy)


REGISTRY["epsilon_greedy"] = EpsilonGreedyActionSelector
