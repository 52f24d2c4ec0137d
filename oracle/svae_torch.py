"""PyTorch-CPU restatement of the notebook's per-user training step -- TEST INFRASTRUCTURE ONLY.

Used for (a) bench.py's ``cpu_baseline`` leg ("port": the notebook's PyTorch CPU path timed on the
GPU box's host cores) and (b) cross-checks in tests.  It runs the step the way the notebook does:
one user per step (nb:80), dense one-hot targets built by per-timestep scatter (nb:500-513), fp32,
torch autograd, torch.optim.Adam(weight_decay=5e-3) (nb:1260-1263).  Pinned by
tests/test_oracle_golden.py::test_torch_port_matches_reference against the golden vectors.
"""
import time

import numpy as np
import torch
import torch.nn.functional as F


class TorchSVAE(torch.nn.Module):
    def __init__(self, total_items, embed=256, rnn=200, hidden=150, latent=64):
        super().__init__()
        nn = torch.nn
        self.latent = latent
        self.item_embed = nn.Embedding(total_items, embed)                 # nb:722
        self.gru = nn.GRU(embed, rnn, batch_first=True, num_layers=1)      # nb:724-727
        self.enc = nn.Linear(rnn, hidden)                                  # Encoder nb:686-688
        self.lat = nn.Linear(hidden, 2 * latent)                           # nb:729
        self.dec1 = nn.Linear(latent, hidden)                              # Decoder nb:701
        self.dec2 = nn.Linear(hidden, total_items)                         # nb:702

    NAME_MAP = {
        "encoder.linear1.weight": "enc.weight", "encoder.linear1.bias": "enc.bias",
        "decoder.linear1.weight": "dec1.weight", "decoder.linear1.bias": "dec1.bias",
        "decoder.linear2.weight": "dec2.weight", "decoder.linear2.bias": "dec2.bias",
        "item_embed.weight": "item_embed.weight",
        "gru.weight_ih_l0": "gru.weight_ih_l0", "gru.weight_hh_l0": "gru.weight_hh_l0",
        "gru.bias_ih_l0": "gru.bias_ih_l0", "gru.bias_hh_l0": "gru.bias_hh_l0",
        "linear1.weight": "lat.weight", "linear1.bias": "lat.bias",
    }

    def load_reference_state(self, sd):
        self.load_state_dict({self.NAME_MAP[k]: torch.as_tensor(np.asarray(v)) for k, v in sd.items()})

    def reference_state(self):
        inv = {v: k for k, v in self.NAME_MAP.items()}
        return {inv[k]: v.detach().numpy().copy() for k, v in self.state_dict().items()}

    def forward(self, x, eps):
        T = x.shape[1]
        e = self.item_embed(x.view(-1)).view(1, T, -1)
        h, _ = self.gru(e)
        h = torch.tanh(self.enc(h.view(T, -1)))
        t = self.lat(h)
        mu, ls = t[:, :self.latent], t[:, self.latent:]
        z = mu + torch.exp(ls) * eps
        out = self.dec2(torch.tanh(self.dec1(z)))
        return out.view(1, T, -1), mu, ls


def vae_loss(dec_out, mu, ls, y, anneal):
    kld = torch.mean(torch.sum(0.5 * (-ls + torch.exp(ls) + mu ** 2 - 1), -1))          # nb:797
    logp = F.log_softmax(dec_out, -1)                                                   # nb:802
    num_ones = float(torch.sum(y[0, 0]))                                                # nb:803
    lik = torch.sum(-1.0 * y.view(-1, y.shape[-1]) * logp.view(-1, y.shape[-1])) / num_ones
    return anneal * kld + lik


def dense_targets(items, next_k, total_items):
    T = len(items) - 1
    y = torch.zeros(1, T, total_items)
    for t in range(T):                                                                  # nb:510-513
        y[0, t].scatter_(0, torch.as_tensor(np.asarray(items[t + 1:t + 1 + next_k], dtype=np.int64)), 1.0)
    return y


def train_users(model, opt, users, next_k, total_items, eps_list=None, anneal_fn=None, rng=None):
    """One notebook-style step per user.  Returns list of losses."""
    losses = []
    for b, items in enumerate(users, start=1):
        x = torch.as_tensor(np.asarray(items[:-1], dtype=np.int64))[None]
        y = dense_targets(items, next_k, total_items)
        T = x.shape[1]
        if eps_list is not None:
            eps = torch.as_tensor(eps_list[b - 1])
        else:
            eps = torch.from_numpy((rng or np.random).normal(0, 1, size=(T, model.latent))).float()   # nb:744
        anneal = anneal_fn(b) if anneal_fn else 0.0
        opt.zero_grad()
        dec, mu, ls = model(x, eps)
        loss = vae_loss(dec, mu, ls, y, anneal)
        loss.backward()
        opt.step()
        losses.append(float(loss.item()))
    return losses


def time_cpu_baseline(users, total_items, next_k=4, warmup=3, budget_s=15.0, max_users=200, seed=0):
    """Sequences/s of the notebook-style CPU step on this host.  Bounded by budget_s."""
    torch.manual_seed(seed)
    model = TorchSVAE(total_items)
    opt = torch.optim.Adam(model.parameters(), weight_decay=5e-3)
    rng = np.random.RandomState(seed)
    train_users(model, opt, users[:warmup], next_k, total_items, rng=rng)
    n, t0 = 0, time.perf_counter()
    for items in users[warmup:warmup + max_users]:
        train_users(model, opt, [items], next_k, total_items, rng=rng)
        n += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return dict(value=n / dt, users=n, seconds=dt, cores=torch.get_num_threads())
