"""Planner interface of the reference (src/planners/planner.py): a config bag and the reset / rollout protocol."""


class PlannerConfig(object):
    def __init__(self, **kwargs):
        for key, val in kwargs.items():
            setattr(self, key, val)


class PlannerNusc(object):
    def __init__(self, map_env, cfg):
        self.map_env = map_env
        self.cfg = cfg

    def reset(self, init_state, vehicle_atts):
        raise NotImplementedError

    def rollout(self, agent_obs, num_steps, init_state=None):
        raise NotImplementedError
