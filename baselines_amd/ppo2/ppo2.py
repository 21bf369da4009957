"""``learn`` of the fork's PPO2 driver (baselines/ppo2/ppo2.py:15-188) on the HIP path.

Reference-literal by default: ``nminibatches`` is the minibatch SIZE (:33,:76), the
advantage is the fork's 5-channel scalarisation (:92-95), lr/cliprange are floored (:82-83).
``upstream=True`` switches to the upstream meaning of the same knobs (``nminibatches`` =
count, ``advs = returns - values``), which is what policies with the upstream constructor
signature need.
"""
import gc
import os
import os.path as osp
import time
from collections import deque

import numpy as np

from .. import hip, logger
from .common_functions import *   # noqa: F401,F403  (Model, Runner, sf01, constfn, compute_advs, REWARD_NUM)
from .common_functions import Model, Runner, compute_advs, constfn


def num_updates(total_timesteps, nbatch):
    """``nupdates = total_timesteps//nbatch`` (baselines/ppo2/ppo2.py:69)."""
    return int(total_timesteps) // nbatch


def update_schedule(update, nupdates, lr, cliprange, upstream=False):
    """(frac, lrnow, cliprangenow) of update number ``update`` (1-based), baselines/ppo2/ppo2.py:78-83:
    ``frac = 1 - (update-1)/nupdates``; the fork multiplies the schedule's value by ``frac`` once more
    (``lrnow = frac*lr(frac)``, :79) and floors the two at 1e-4 and 0.1 (``np.clip(.., 1e-4, 1)``, :82;
    ``np.clip(.., 0.1, 1)``, :83).  ``upstream``: the schedule's own values, unclipped."""
    frac = 1.0 - (update - 1.0) / nupdates
    if upstream:
        return frac, lr(frac), cliprange(frac)
    return frac, np.clip(frac * lr(frac), 1e-4, 1), np.clip(cliprange(frac), 0.1, 1)


def update_fps(nbatch, seconds):
    """``fps = int(nbatch / (tnow - tstart))`` (baselines/ppo2/ppo2.py:141)."""
    return int(nbatch / seconds)


REFERENCE_CHECKPOINT_DIR = osp.join('../model', 'checkpoints')   # baselines/ppo2/ppo2.py:178


def learn(*, policy, env, nsteps, total_timesteps, ent_coef, lr,
          vf_coef=0.5, max_grad_norm=0.5, gamma=0.99, lam=0.95,
          log_interval=10, nminibatches=4, noptepochs=4, cliprange=0.2,
          save_interval=0, upstream=None, comm=None, device_path=True, return_model=False,
          checkpoint_dir=None, reference_rng_parity=False):
    """Build-only keyword arguments beyond the reference's: ``upstream``, ``comm`` (a DataParallel context),
    ``device_path`` (False = the reference's host minibatch loop through ``Model.train``), ``return_model``,
    ``checkpoint_dir`` (None = the reference's literal ``../model/checkpoints``, :178), and
    ``reference_rng_parity``: the fork also builds the never-used ``model_pf`` and ``model_pf_2`` (:45-46),
    whose ``ortho_init`` draws advance the global ``np.random`` stream before the first shuffle; True
    consumes the same draws so that a seeded run's index stream lines up with the fork's end to end."""
    if isinstance(lr, float): lr = constfn(lr)
    else: assert callable(lr)
    if isinstance(cliprange, float): cliprange = constfn(cliprange)
    else: assert callable(cliprange)
    total_timesteps = int(total_timesteps)
    if upstream is None:
        upstream = getattr(policy, "signature", "fork") != "fork"

    nenvs = env.num_envs
    ob_space = env.observation_space
    ac_space = env.action_space
    nbatch = nenvs * nsteps
    nbatch_train = nbatch // nminibatches if upstream else nminibatches   # :31 (upstream) / :33 (fork)

    make_model = lambda model_name, need_summary: Model(
        model_name=model_name, policy=policy, ob_space=ob_space, ac_space=ac_space, nbatch_act=nenvs,
        nbatch_train=nbatch_train, nsteps=nsteps, ent_coef=ent_coef, vf_coef=vf_coef, lr=lr(1),
        max_grad_norm=max_grad_norm, need_summary=need_summary, upstream_loss=upstream, comm=comm)
    model = make_model('model', need_summary=True)
    # The fork also builds 'model_pf' and 'model_pf_2' (:45-46); the live code never uses them.  They are not
    # kept here; with reference_rng_parity their initialisers' draws are consumed from np.random all the same.
    if reference_rng_parity:
        for extra in ('model_pf', 'model_pf_2'):
            make_model(extra, need_summary=False)

    runner = Runner(env=env, model=model, nsteps=nsteps, gamma=gamma, lam=lam)
    epinfobuf = deque(maxlen=100)
    tfirststart = time.time()
    nupdates = num_updates(total_timesteps, nbatch)
    max_return = [200, 120, 0, 0, 0]   # :71

    # Everything built so far (torch, the model, the runner) is long-lived: taken out of the cyclic collector's view, a
    # full collection inside the loop walks the loop's own few objects instead of pausing the thread that feeds the GPU
    # for 35-48 ms (thirty updates of the MLP configurations).
    gc.collect()
    gc.freeze()

    for update in range(1, nupdates + 1):
        assert nbatch % nminibatches == 0   # :74
        tstart = time.time()
        frac, lrnow, cliprangenow = update_schedule(update, nupdates, lr, cliprange, upstream)   # :78-83
        obs, returns, masks, actions, values, rewards, neglogpacs, states, epinfos, ret = runner.run(is_test=False)
        epinfobuf.extend(epinfos)
        ro = runner.rollout
        if upstream:
            ro.advs = ro.advs_ori.reshape(ro.T, ro.E, -1)[:, :, 0].contiguous() if ro.advs_ori.shape[2] != 1 \
                else ro.advs_ori.reshape(ro.T, ro.E)
        else:
            # advs_ori = returns - values; advs = compute_advs(max_return, values, advs_ori, [], [])  (:92-95)
            N, R = ro.T * ro.E, ro.values.shape[2]
            ro.advs = compute_advs(max_return[:R], ro.values.reshape(N, R), ro.advs_ori.reshape(N, R), [], []) \
                .reshape(ro.T, ro.E)
        if rewards is None:                       # device-resident env: the rollout never left HBM
            assert device_path, "an env that lives on the device needs device_path=True"
            rewards = ro.rewards.mean().item() if (update % log_interval == 0 or update == 1) else 0.0
        if device_path:
            lossvals = model.train_on_rollout(float(lrnow), float(cliprangenow), ro, noptepochs, nbatch_train,
                                               more_updates=update < nupdates).cpu().numpy()
            model.check_faults()                  # a stale index list raises here, as NumPy's fancy indexing would
        else:
            # the reference's host loop (:109-137): shuffle, fancy-index gather, Model.train per minibatch
            advs_host = sf01_host(ro.advs)
            mblossvals = []
            if states is None:
                inds = np.arange(nbatch)
                for _ in range(noptepochs):
                    np.random.shuffle(inds)
                    for start in range(0, nbatch, nbatch_train):
                        mbinds = inds[start:start + nbatch_train]
                        if upstream:
                            slices = (arr[mbinds] for arr in (obs, returns, masks, actions, values, neglogpacs))
                        else:
                            slices = (arr[mbinds] for arr in (obs, returns, masks, actions, values, advs_host, neglogpacs))
                        mblossvals.append(model.train(float(lrnow), float(cliprangenow), *slices))
            else:
                envinds = np.arange(nenvs)
                flatinds = np.arange(nenvs * nsteps).reshape(nenvs, nsteps)
                envsperbatch = nbatch_train // nsteps
                for _ in range(noptepochs):
                    np.random.shuffle(envinds)
                    for start in range(0, nenvs, envsperbatch):
                        mbenvinds = envinds[start:start + envsperbatch]
                        mbflatinds = flatinds[mbenvinds].ravel()
                        slices = (arr[mbflatinds] for arr in (obs, returns, masks, actions, values, neglogpacs))
                        mblossvals.append(model.train(float(lrnow), float(cliprangenow), *slices, states[mbenvinds]))
            lossvals = np.mean(mblossvals, axis=0)
        tnow = time.time()
        fps = update_fps(nbatch, tnow - tstart)
        if update % log_interval == 0 or update == 1:
            logger.logkv("serial_timesteps", update * nsteps)
            logger.logkv("nupdates", update)
            logger.logkv("total_timesteps", update * nbatch)
            logger.logkv("fps", fps)
            logger.logkv('rew_mean', float(np.mean(rewards)))
            logger.logkv('eprewmean', safemean([np.asarray(epinfo['r']).ravel()[-1] for epinfo in epinfobuf]))
            logger.logkv('eplenmean', safemean([epinfo['l'] for epinfo in epinfobuf]))
            logger.logkv('time_elapsed', tnow - tfirststart)
            for (lossval, lossname) in zip(lossvals, model.loss_names):
                logger.logkv(lossname, float(lossval))
            logger.dumpkvs()
        if save_interval and (update % save_interval == 0 or update == 1 or update == nupdates) and logger.get_dir():
            checkdir = REFERENCE_CHECKPOINT_DIR if checkpoint_dir is None else checkpoint_dir
            os.makedirs(checkdir, exist_ok=True)
            model.save(osp.join(checkdir, '%.5i' % update), osp.join(checkdir, '%.5i' % 0))
    gc.unfreeze()
    env.close()
    if return_model:
        return model


def sf01_host(dev_tm):
    """Device [T, E] -> host env-major flat [E*T]."""
    from .. import hip
    return hip.sf01(dev_tm.contiguous()).cpu().numpy()


def safemean(xs):
    return np.nan if len(xs) == 0 else np.mean(xs)
