"""Abstract multi-agent env interface (reference: src/envs/multiagentenv.py:6-75)."""


class MultiAgentEnv:
    def step(self, actions):
        raise NotImplementedError

    def get_obs(self):
        raise NotImplementedError

    def get_obs_agent(self, agent_id):
        raise NotImplementedError

    def get_obs_size(self):
        raise NotImplementedError

    def get_state(self):
        raise NotImplementedError

    def get_state_size(self):
        raise NotImplementedError

    def get_avail_actions(self):
        raise NotImplementedError

    def get_avail_agent_actions(self, agent_id):
        raise NotImplementedError

    def get_total_actions(self):
        raise NotImplementedError

    def reset(self):
        raise NotImplementedError

    def render(self):
        raise NotImplementedError

    def close(self):
        raise NotImplementedError

    def seed(self):
        raise NotImplementedError

    def save_replay(self):
        raise NotImplementedError

    def get_own_feature_size(self):
        return None

    def get_units_type_id(self):
        return None

    def get_env_info(self):
        return {"state_shape": self.get_state_size(), "obs_shape": self.get_obs_size(), "n_actions": self.get_total_actions(),
                "n_agents": self.n_agents, "episode_limit": self.episode_limit, "units_type_id": self.get_units_type_id(),
                "own_feature_size": self.get_own_feature_size()}
