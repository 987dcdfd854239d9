"""Oracle restatement of the per-env closures compiled into the reference graph.

Each env in the reference supplies obs_preproc / obs_postproc / targ_proc /
tf_reward_fn.  Restated in numpy, dtype-preserving.
Test infrastructure only (see oracle/__init__.py).

Reference call sites (under /root/reference/cadm/envs/):
  half_cheetah_env.py:46-59 (preproc/postproc/targ_proc), :82-88 (tf_reward_fn)
  half_cheetah_cripple_env.py:60-73, :90-96      (identical closures)
  ant_env.py:52-62, :89-98
  slim_humanoid_env.py:39-46, :95-111
  classic_control.py:94-101,154-166 (cartpole), :209-218,284-291 (pendulum)
"""
import math

import numpy as np


class EnvSpec:
    """Shape + closure bundle for one env kind (no simulator)."""

    name = None
    obs_dim = None
    act_dim = None
    proc_obs_dim = None
    discrete = False

    def obs_preproc(self, obs):
        raise NotImplementedError

    def obs_postproc(self, obs, pred):
        raise NotImplementedError

    def targ_proc(self, obs, next_obs):
        raise NotImplementedError

    def reward(self, obs, act, next_obs):
        """act is [..., A] continuous, or the raw action for discrete envs."""
        raise NotImplementedError


class HalfCheetah(EnvSpec):
    # half_cheetah_env.py:39-44: obs = [vel(1), qpos[1:](8), qvel(9)] -> 18
    name = "halfcheetah"
    obs_dim, act_dim, proc_obs_dim = 18, 6, 18

    def obs_preproc(self, obs):  # half_cheetah_env.py:46-50
        return np.concatenate([obs[..., 1:2], np.sin(obs[..., 2:3]),
                               np.cos(obs[..., 2:3]), obs[..., 3:]], axis=-1)

    def obs_postproc(self, obs, pred):  # half_cheetah_env.py:52-56
        return np.concatenate([pred[..., :1], obs[..., 1:] + pred[..., 1:]], axis=-1)

    def targ_proc(self, obs, next_obs):  # half_cheetah_env.py:58-59
        return np.concatenate([next_obs[..., :1], next_obs[..., 1:] - obs[..., 1:]], axis=-1)

    def reward(self, obs, act, next_obs):  # half_cheetah_env.py:82-88 (pre-step obs)
        dt = obs.dtype.type
        ctrl_cost = dt(1e-1) * np.sum(np.square(act), axis=-1)
        return obs[..., 0] - ctrl_cost


class CrippleHalfCheetah(HalfCheetah):
    # half_cheetah_cripple_env.py:60-73,90-96: same closures as halfcheetah
    name = "cripple_halfcheetah"


class Ant(EnvSpec):
    # ant_env.py:45-50: obs = [vel(1), qpos[2:](13), qvel(14)] -> 28
    name = "ant"
    obs_dim, act_dim, proc_obs_dim = 28, 8, 27

    def obs_preproc(self, obs):  # ant_env.py:52-53
        return obs[..., 1:]

    def obs_postproc(self, obs, pred):  # ant_env.py:55-59
        return np.concatenate([pred[..., :1], obs[..., 1:] + pred[..., 1:]], axis=-1)

    def targ_proc(self, obs, next_obs):  # ant_env.py:61-62
        return np.concatenate([next_obs[..., :1], next_obs[..., 1:] - obs[..., 1:]], axis=-1)

    def reward(self, obs, act, next_obs):  # ant_env.py:89-98
        dt = obs.dtype.type
        reward_ctrl = dt(-0.005) * np.sum(np.square(act), axis=-1)
        reward_run = obs[..., 0]
        # reference order: reward_run + reward_ctrl + reward_contact(0.0) + reward_survive(0.05)
        return reward_run + reward_ctrl + dt(0.0) + dt(0.05)


class SlimHumanoid(EnvSpec):
    # slim_humanoid_env.py:34-37: obs = [qpos[2:](22), qvel(23)] -> 45
    name = "slim_humanoid"
    obs_dim, act_dim, proc_obs_dim = 45, 17, 45

    def obs_preproc(self, obs):  # slim_humanoid_env.py:39-40
        return obs

    def obs_postproc(self, obs, pred):  # slim_humanoid_env.py:42-43
        return obs + pred

    def targ_proc(self, obs, next_obs):  # slim_humanoid_env.py:45-46
        return next_obs - obs

    def reward(self, obs, act, next_obs):  # slim_humanoid_env.py:95-111 (tf_reward_fn)
        dt = obs.dtype.type
        lin_vel_cost = dt(0.25 / 0.015) * obs[..., 22]
        quad_ctrl_cost = dt(0.1) * np.sum(np.square(act), axis=-1)
        alive = np.logical_and(obs[..., 1] > dt(1.0), obs[..., 1] < dt(2.0))
        alive_bonus = dt(5.0) * alive.astype(obs.dtype)
        return lin_vel_cost - quad_ctrl_cost - dt(0.0) + alive_bonus


class CartPole(EnvSpec):
    # classic_control.py:94-101 closures; :154-166 tf_reward_fn (uses NEXT obs)
    name = "cartpole"
    obs_dim, act_dim, proc_obs_dim = 4, 2, 4
    discrete = True

    def obs_preproc(self, obs):
        return obs

    def obs_postproc(self, obs, pred):
        return obs + pred

    def targ_proc(self, obs, next_obs):
        return next_obs - obs

    def reward(self, obs, act, next_obs):
        dt = next_obs.dtype.type
        x_threshold = dt(2.4)
        theta_threshold_radians = dt(12 * 2 * math.pi / 360)
        cond = ((next_obs[..., 0] > x_threshold).astype(next_obs.dtype)
                + (next_obs[..., 0] < -x_threshold).astype(next_obs.dtype)
                + (next_obs[..., 2] > theta_threshold_radians).astype(next_obs.dtype)
                + (next_obs[..., 2] < -theta_threshold_radians).astype(next_obs.dtype))
        return dt(1) - cond * dt(1)


class Pendulum(EnvSpec):
    # classic_control.py:284-291 closures; :209-218 tf_reward_fn; max_torque=2.0 (:172 region)
    name = "pendulum"
    obs_dim, act_dim, proc_obs_dim = 3, 1, 3
    max_torque = 2.0

    def obs_preproc(self, obs):
        return obs

    def obs_postproc(self, obs, pred):
        return obs + pred

    def targ_proc(self, obs, next_obs):
        return next_obs - obs

    def reward(self, obs, act, next_obs):
        dt = obs.dtype.type
        theta = np.arctan2(obs[..., 1], obs[..., 0])
        # python/TF floormod semantics (result has the divisor's sign)
        theta_normalize = np.mod(theta + dt(np.pi), dt(2 * np.pi)) - dt(np.pi)
        thetadot = obs[..., 2]
        torque = np.clip(act, dt(-self.max_torque), dt(self.max_torque))
        torque = torque.reshape(torque.shape[:-1])
        cost = theta_normalize ** 2 + dt(0.1) * thetadot ** 2 + dt(0.001) * torque ** 2
        return -cost


ENVS = {cls.name: cls for cls in
        (HalfCheetah, CrippleHalfCheetah, Ant, SlimHumanoid, CartPole, Pendulum)}


def make_env(name):
    return ENVS[name]()
