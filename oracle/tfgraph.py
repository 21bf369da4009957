"""The TensorFlow-graph half of the reference, restated on torch-CPU.

Test infrastructure (see ``oracle/__init__.py``).  PARITY UNPINNED: the
reference evaluates these graphs inside TensorFlow 1.x, which is not available
and for which the reference holds no golden vectors.  Every function follows
the cited graph op by op; gradients come from ``torch.autograd`` (float64 by
default) and are cross-checked against finite differences in the tests.

Parameters travel as an ordered dict name -> tensor.  Names are the reference's
variable-scope names (``pi_fc1/w`` ...), order is creation order, which is also
the checkpoint order of ``Model.save`` (``baselines/ppo2/common_functions.py:115-119``).
"""
import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

from .init import ortho_init

LOG_2PI = math.log(2.0 * math.pi)


# ----------------------------------------------------------------------------
# parameter construction
# ----------------------------------------------------------------------------

def mlp_param_shapes(ob_dim, hidden, ac_dim, nvalue, gaussian, order="fork"):
    """Shapes of the two-trunk MLP policy in creation order.

    order="fork":     ``baselines/ppo2/policies.py:139-149`` (actor scope then value scope)
    order="upstream": ``baselines/a2c/policies.py:123-129`` (+ ``pdfromlatent``,
                      ``baselines/common/distributions.py:54-56,86-90``)
    """
    h1, h2 = hidden
    actor = [("pi_fc1/w", (ob_dim, h1)), ("pi_fc1/b", (h1,)),
             ("pi_fc2/w", (h1, h2)), ("pi_fc2/b", (h2,))]
    head = [("pi/w", (h2, ac_dim)), ("pi/b", (ac_dim,))]
    if gaussian:
        head.append(("logstd", (1, ac_dim)))
    value = [("vf_fc1/w", (ob_dim, h1)), ("vf_fc1/b", (h1,)),
             ("vf_fc2/w", (h1, h2)), ("vf_fc2/b", (h2,)),
             ("vf/w", (h2, nvalue)), ("vf/b", (nvalue,))]
    if order == "fork":
        return OrderedDict(actor + head + value)
    return OrderedDict(actor + value + head)


def cnn_param_shapes(nc, nact, lstm=None, nlstm=256, flat=3136):
    """``nature_cnn`` + heads, ``baselines/ppo2/policies.py:6-16,58-130``;
    ``lstm`` in (None, "lstm", "lnlstm") inserts ``baselines/a2c/utils.py:86-147``."""
    shapes = [("c1/w", (8, 8, nc, 32)), ("c1/b", (32,)),
              ("c2/w", (4, 4, 32, 64)), ("c2/b", (64,)),
              ("c3/w", (3, 3, 64, 64)), ("c3/b", (64,)),
              ("fc1/w", (flat, 512)), ("fc1/b", (512,))]
    feat = 512
    if lstm == "lstm":
        shapes += [("lstm1/wx", (512, 4 * nlstm)), ("lstm1/wh", (nlstm, 4 * nlstm)),
                   ("lstm1/b", (4 * nlstm,))]
        feat = nlstm
    elif lstm == "lnlstm":
        shapes += [("lstm1/wx", (512, 4 * nlstm)), ("lstm1/gx", (4 * nlstm,)), ("lstm1/bx", (4 * nlstm,)),
                   ("lstm1/wh", (nlstm, 4 * nlstm)), ("lstm1/gh", (4 * nlstm,)), ("lstm1/bh", (4 * nlstm,)),
                   ("lstm1/b", (4 * nlstm,)), ("lstm1/gc", (nlstm,)), ("lstm1/bc", (nlstm,))]
        feat = nlstm
    shapes += [("pi/w", (feat, nact)), ("pi/b", (nact,)), ("v/w", (feat, 1)), ("v/b", (1,))]
    return OrderedDict(shapes)


_INIT_SCALE = {"pi/w": 0.01}


def init_params(shapes, rng, pi_scale=0.01, head_scale=1.0):
    """Initial values as the layer builders give them: ``ortho_init(sqrt 2)`` for
    trunk weights, ``pi_scale`` for the policy head, 1.0 for value heads and LSTM
    matrices, zeros for biases/logstd, ones for layer-norm gains
    (``baselines/a2c/utils.py:57-58,66-67,90-92,120-131``)."""
    out = OrderedDict()
    for name, shp in shapes.items():
        leaf = name.split("/")[-1]
        if leaf in ("w", "wx", "wh"):
            if name == "pi/w":
                scale = pi_scale
            elif name in ("vf/w", "v/w") or leaf in ("wx", "wh"):
                scale = head_scale
            else:
                scale = math.sqrt(2.0)
            out[name] = ortho_init(shp, scale, rng)
        elif leaf in ("gx", "gh", "gc"):
            out[name] = np.ones(shp, np.float32)
        else:
            out[name] = np.zeros(shp, np.float32)
    return out


def to_torch(params, dtype=torch.float64, requires_grad=True):
    out = OrderedDict()
    for k, v in params.items():
        t = torch.as_tensor(np.asarray(v)).to(dtype).clone()
        t.requires_grad_(requires_grad)
        out[k] = t
    return out


# ----------------------------------------------------------------------------
# networks
# ----------------------------------------------------------------------------

def fc(x, p, scope):
    """``baselines/a2c/utils.py:63-68``"""
    return x @ p[scope + "/w"] + p[scope + "/b"]


def mlp_forward(p, obs):
    """Two separate tanh trunks; returns (pi, vf[B, nvalue]).
    ``baselines/ppo2/policies.py:139-149`` / ``baselines/a2c/policies.py:123-129``."""
    x = obs
    h = torch.tanh(fc(x, p, "pi_fc1"))
    h = torch.tanh(fc(h, p, "pi_fc2"))
    pi = fc(h, p, "pi")
    g = torch.tanh(fc(x, p, "vf_fc1"))
    g = torch.tanh(fc(g, p, "vf_fc2"))
    vf = fc(g, p, "vf")
    return pi, vf


def conv(x_nhwc, p, scope, stride):
    """VALID NHWC convolution with ``[rf, rf, cin, cout]`` weights,
    ``baselines/a2c/utils.py:42-61``."""
    w = p[scope + "/w"].permute(3, 2, 0, 1)
    y = F.conv2d(x_nhwc.permute(0, 3, 1, 2), w, bias=p[scope + "/b"].reshape(-1), stride=stride)
    return y.permute(0, 2, 3, 1)


def nature_cnn(p, obs_u8, relu_on=None, acts_out=None, pre_out=None):
    """``baselines/ppo2/policies.py:6-16``; flatten keeps NHWC order (``conv_to_fc``
    ``baselines/a2c/utils.py:149-152``).

    ``relu_on``: optional four boolean arrays (c1, c2, c3 outputs in NHWC, fc1 output) that say which units
    are switched on; the rectifiers then multiply by them instead of testing their own input's sign.  A float32
    implementation and this float64 one disagree about the sign of the few pre-activations that lie within
    float32 rounding of zero (a handful among 10^7), and each such unit opens or closes a whole gradient
    path; with the implementation's own on/off pattern handed in, the two compute the same function.
    ``acts_out``: optional list that receives the four activations; ``pre_out``: one that receives the four
    pre-activations (the rectifiers' inputs)."""
    dtype = p["c1/w"].dtype
    x = obs_u8.to(dtype) / 255.0

    def rect(z, i):
        if pre_out is not None:
            pre_out.append(z)
        h = F.relu(z) if relu_on is None else z * torch.as_tensor(relu_on[i]).reshape(z.shape).to(dtype)
        if acts_out is not None:
            acts_out.append(h)
        return h
    h = rect(conv(x, p, "c1", 4), 0)
    h = rect(conv(h, p, "c2", 2), 1)
    h = rect(conv(h, p, "c3", 1), 2)
    h = h.reshape(h.shape[0], -1)
    return rect(fc(h, p, "fc1"), 3)


def cnn_forward(p, obs_u8, relu_on=None, acts_out=None, pre_out=None):
    """``CnnPolicy``, ``baselines/ppo2/policies.py:100-130``; returns (logits, vf[B,1])."""
    h = nature_cnn(p, obs_u8, relu_on, acts_out, pre_out)
    return fc(h, p, "pi"), fc(h, p, "v")


def _ln(x, g, b, e=1e-5):
    """``baselines/a2c/utils.py:110-114`` (moments over axis 1, biased variance)."""
    u = x.mean(dim=1, keepdim=True)
    s = ((x - u) ** 2).mean(dim=1, keepdim=True)
    return (x - u) / torch.sqrt(s + e) * g + b


def lstm_forward(p, xs, masks, state, nenv, nsteps, layer_norm=False):
    """``lstm`` / ``lnlstm``, ``baselines/a2c/utils.py:86-147``.

    xs    : [nenv*nsteps, nin]  env-major rows (``batch_to_seq`` :70-75)
    masks : [nenv*nsteps]       done-before-step flags
    state : [nenv, 2*nh]        ``[c | h]``
    Returns (h rows env-major [nenv*nsteps, nh], new state).
    """
    nh = p["lstm1/wh"].shape[0]
    x = xs.reshape(nenv, nsteps, -1)
    m = masks.reshape(nenv, nsteps)
    c, h = state[:, :nh], state[:, nh:]
    outs = []
    for t in range(nsteps):
        mt = m[:, t:t + 1]
        c = c * (1 - mt)
        h = h * (1 - mt)
        if layer_norm:
            z = (_ln(x[:, t] @ p["lstm1/wx"], p["lstm1/gx"], p["lstm1/bx"])
                 + _ln(h @ p["lstm1/wh"], p["lstm1/gh"], p["lstm1/bh"]) + p["lstm1/b"])
        else:
            z = x[:, t] @ p["lstm1/wx"] + h @ p["lstm1/wh"] + p["lstm1/b"]
        i, f, o, u = torch.split(z, nh, dim=1)
        i, f, o, u = torch.sigmoid(i), torch.sigmoid(f), torch.sigmoid(o), torch.tanh(u)
        c = f * c + i * u
        if layer_norm:
            h = o * torch.tanh(_ln(c, p["lstm1/gc"], p["lstm1/bc"]))
        else:
            h = o * torch.tanh(c)
        outs.append(h)
    hs = torch.stack(outs, dim=1).reshape(nenv * nsteps, nh)
    return hs, torch.cat([c, h], dim=1)


def cnn_lstm_forward(p, obs_u8, masks, state, nenv, nsteps, layer_norm=False, relu_on=None, acts_out=None, pre_out=None):
    """``LstmPolicy`` / ``LnLstmPolicy``, ``baselines/ppo2/policies.py:18-98``."""
    h = nature_cnn(p, obs_u8, relu_on, acts_out, pre_out)
    h5, snew = lstm_forward(p, h, masks, state, nenv, nsteps, layer_norm)
    return fc(h5, p, "pi"), fc(h5, p, "v"), snew


# ----------------------------------------------------------------------------
# distributions
# ----------------------------------------------------------------------------

def gaussian_neglogp(mean, logstd, x):
    """``DiagGaussianPd.neglogp``, ``baselines/common/distributions.py:202-205``;
    ``logstd`` is the ``[1, A]`` parameter broadcast as ``mean*0 + logstd``
    (``baselines/ppo2/policies.py:151``)."""
    ls = mean * 0.0 + logstd
    std = torch.exp(ls)
    return (0.5 * torch.sum(((x - mean) / std) ** 2, dim=-1)
            + 0.5 * LOG_2PI * x.shape[-1] + torch.sum(ls, dim=-1))


def gaussian_entropy(mean, logstd):
    """``baselines/common/distributions.py:209-210``"""
    ls = mean * 0.0 + logstd
    return torch.sum(ls + 0.5 * math.log(2.0 * math.pi * math.e), dim=-1)


def categorical_neglogp(logits, x):
    """softmax cross-entropy against the one-hot action,
    ``baselines/common/distributions.py:141-148``."""
    return F.cross_entropy(logits, x.long(), reduction="none")


def categorical_entropy(logits):
    """``baselines/common/distributions.py:158-163``"""
    a0 = logits - logits.max(dim=-1, keepdim=True).values
    ea0 = torch.exp(a0)
    z0 = ea0.sum(dim=-1, keepdim=True)
    p0 = ea0 / z0
    return torch.sum(p0 * (torch.log(z0) - a0), dim=-1)


# ----------------------------------------------------------------------------
# loss, clip, Adam
# ----------------------------------------------------------------------------

def tf_maximum(a, b):
    """``tf.maximum`` with TF's tie rule: the gradient goes to ``a`` where ``a >= b``."""
    return torch.where(a >= b, a, b)


class LossCfg:
    """Which of the two loss/optimiser variants of
    ``baselines/ppo2/common_functions.py`` is in force.

    fork (live code :26-54):       vf_loss = mean((v-R)^2) over [B, R]; actor and value
                                   objectives minimised separately (equivalent to one
                                   elementwise Adam over both parameter sets), no clipping.
    upstream (commented :30-33,:56-63): vf_loss = .5*mean(max((v-R)^2, (vclip-R)^2));
                                   loss = pg - ent_coef*H + vf_coef*vf; clip_by_global_norm.
    """

    def __init__(self, ent_coef=0.0, vf_coef=0.5, max_grad_norm=0.5, upstream=True):
        self.ent_coef = ent_coef
        self.vf_coef = vf_coef
        self.max_grad_norm = max_grad_norm
        self.upstream = upstream


def ppo_losses(neglogpac, entropy_rows, vpred, batch, cliprange, cfg):
    """Loss graph, ``baselines/ppo2/common_functions.py:26-44``.

    batch: dict with ADV [B] (already normalised), R [B, nvalue], OLDNEGLOGPAC [B],
    OLDVPRED [B, nvalue].  Returns (total objective for autograd, 5 reported scalars).
    """
    ADV, R, OLDNLP, OLDV = batch["ADV"], batch["R"], batch["OLDNEGLOGPAC"], batch["OLDVPRED"]
    entropy = entropy_rows.mean()
    vf_losses1 = (vpred - R) ** 2
    if cfg.upstream:
        vpredclipped = OLDV + torch.clamp(vpred - OLDV, -cliprange, cliprange)
        vf_losses2 = (vpredclipped - R) ** 2
        vf_loss = 0.5 * tf_maximum(vf_losses1, vf_losses2).mean()
    else:
        vf_loss = vf_losses1.mean()
    ratio = torch.exp(OLDNLP - neglogpac)
    pg_losses = -ADV * ratio
    pg_losses2 = -ADV * torch.clamp(ratio, 1.0 - cliprange, 1.0 + cliprange)
    pg_loss = tf_maximum(pg_losses, pg_losses2).mean()
    approxkl = 0.5 * ((neglogpac - OLDNLP) ** 2).mean()
    clipfrac = (torch.abs(ratio - 1.0) > cliprange).to(ratio.dtype).mean()
    if cfg.upstream:
        total = pg_loss - entropy * cfg.ent_coef + vf_loss * cfg.vf_coef
    else:
        # two objectives over disjoint parameter sets == their sum for the gradient
        total = (pg_loss - entropy * cfg.ent_coef) + vf_loss
    return total, (pg_loss, vf_loss, entropy, approxkl, clipfrac)


def clip_by_global_norm(grads, clip_norm):
    """TF ``clip_by_global_norm`` as used at ``baselines/ppo2/common_functions.py:59-60``
    and live at ``baselines/a2c/a2c.py:40-42``: ``g * c / max(||g||, c)``."""
    gn = torch.sqrt(sum((g.double() ** 2).sum() for g in grads))
    scale = clip_norm / max(float(gn), clip_norm)
    return [g * scale for g in grads], float(gn)


class Adam:
    """``tf.train.AdamOptimizer(lr, epsilon=1e-5)`` in the arithmetic of
    ``baselines/common/mpi_adam.py:30-35``: bias correction folded into the step
    size, epsilon added to ``sqrt(v)``."""

    def __init__(self, params, beta1=0.9, beta2=0.999, epsilon=1e-5):
        self.m = [torch.zeros_like(p) for p in params]
        self.v = [torch.zeros_like(p) for p in params]
        self.t = 0
        self.beta1, self.beta2, self.epsilon = beta1, beta2, epsilon

    def update(self, params, grads, stepsize):
        self.t += 1
        a = stepsize * math.sqrt(1 - self.beta2 ** self.t) / (1 - self.beta1 ** self.t)
        with torch.no_grad():
            for i, (p, g) in enumerate(zip(params, grads)):
                self.m[i] = self.beta1 * self.m[i] + (1 - self.beta1) * g
                self.v[i] = self.beta2 * self.v[i] + (1 - self.beta2) * (g * g)
                p += (-a) * self.m[i] / (torch.sqrt(self.v[i]) + self.epsilon)


# ----------------------------------------------------------------------------
# one Model.train call
# ----------------------------------------------------------------------------

class TrainOracle:
    """``Model.train`` (``baselines/ppo2/common_functions.py:65-106``) on the restated graph.

    kind: "mlp" | "cnn" | "lstm" | "lnlstm";  dist: "gaussian" | "categorical".
    Losses are evaluated with the pre-update parameters.
    """

    def __init__(self, params, kind, dist, cfg, dtype=torch.float64, nsteps=None):
        self.p = to_torch(params, dtype)
        self.kind, self.dist, self.cfg, self.dtype = kind, dist, cfg, dtype
        self.nsteps = nsteps
        self.adam = Adam(list(self.p.values()))
        self.last_grads = None
        self.last_grad_norm = None

    def forward(self, obs, masks=None, states=None, relu_on=None):
        p = self.p
        self.last_acts = []     # the rectified activations of the nature_cnn trunk (c1, c2, c3, fc1), when there is one
        self.last_pre = []      # ... and the rectifiers' inputs
        if self.kind == "mlp":
            pi, vf = mlp_forward(p, torch.as_tensor(obs).to(self.dtype))
            snew = None
        elif self.kind == "cnn":
            pi, vf = cnn_forward(p, torch.as_tensor(obs), relu_on, self.last_acts, self.last_pre)
            snew = None
        else:
            nenv = states.shape[0]
            pi, vf, snew = cnn_lstm_forward(
                p, torch.as_tensor(obs), torch.as_tensor(masks).to(self.dtype),
                torch.as_tensor(states).to(self.dtype), nenv, obs.shape[0] // nenv,
                layer_norm=(self.kind == "lnlstm"), relu_on=relu_on, acts_out=self.last_acts, pre_out=self.last_pre)
        return pi, vf, snew

    def neglogp_entropy(self, pi, actions):
        if self.dist == "gaussian":
            a = torch.as_tensor(actions).to(self.dtype)
            return gaussian_neglogp(pi, self.p["logstd"], a), gaussian_entropy(pi, self.p["logstd"])
        return categorical_neglogp(pi, torch.as_tensor(actions)), categorical_entropy(pi)

    def train(self, lr, cliprange, obs, returns, masks, actions, values, advs, neglogpacs,
              states=None, normalize=True, apply=True, relu_on=None):
        """``relu_on``: see :func:`nature_cnn` — the on/off pattern of the rectifiers, taken from the implementation
        under test, so that units within float32 rounding of zero take the same side in both."""
        advs = np.asarray(advs)
        if normalize:
            advs = (advs - advs.mean()) / (advs.std() + 1e-8)
        dt = self.dtype
        nb = obs.shape[0]
        batch = {
            "ADV": torch.as_tensor(advs.astype(np.float32)).to(dt),   # fed to a float32 placeholder (:19)
            "R": torch.as_tensor(np.asarray(returns, np.float32).reshape(nb, -1)).to(dt),
            "OLDNEGLOGPAC": torch.as_tensor(np.asarray(neglogpacs, np.float32)).to(dt),
            "OLDVPRED": torch.as_tensor(np.asarray(values, np.float32).reshape(nb, -1)).to(dt),
        }
        for t in self.p.values():
            t.grad = None
        pi, vf, _ = self.forward(obs, masks, states, relu_on)
        nlp, ent = self.neglogp_entropy(pi, actions)
        total, scalars = ppo_losses(nlp, ent, vf, batch, cliprange, self.cfg)
        plist = list(self.p.values())
        grads = torch.autograd.grad(total, plist, allow_unused=True)
        grads = [g if g is not None else torch.zeros_like(p) for g, p in zip(grads, plist)]
        self.last_grads = OrderedDict((k, g.detach().clone()) for k, g in zip(self.p.keys(), grads))
        if self.cfg.upstream and self.cfg.max_grad_norm is not None:
            grads, self.last_grad_norm = clip_by_global_norm(grads, self.cfg.max_grad_norm)
        if apply:
            self.adam.update(plist, grads, lr)
        return [float(s.detach()) for s in scalars]

    def numpy_params(self):
        return OrderedDict((k, v.detach().cpu().numpy().copy()) for k, v in self.p.items())
