import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import gym_acas2d_amd as g
N, E, T = 64, 640, 40
dtype = torch.float32
gen = torch.Generator(device="cuda:0").manual_seed(11)
actions = torch.rand(T, E, generator=gen, device="cuda:0", dtype=dtype) * 2 - 1
a = g.ACAS2DVecEnv(E, N, device="cuda:0", dtype=dtype, seed=21, env_offset=3)
b = g.ACAS2DVecEnv(E, N, device="cuda:0", dtype=dtype, seed=21, env_offset=3, double_buffer=False)
a.reset(); b.reset()
names = ("own_x", "own_y", "own_psi", "own_v", "goal_x", "goal_y", "trf_x", "trf_y", "trf_psi", "trf_v", "steps", "total_reward", "episode")
for t in range(T):
    oa, ra, da, ia = a.step(actions[t]); ob, rb, db, ib = b.step(actions[t])
    torch.cuda.synchronize()
    bad = False
    for nm, x, y in [("obs", oa, ob), ("rew", ra, rb), ("done", da, db)] + [(n, getattr(a, n), getattr(b, n)) for n in names]:
        x = x.reshape(E, -1); y = y.reshape(E, -1)
        neq = ~((x == y) | ((x != x) & (y != y)))
        if neq.any():
            bad = True
            e = neq.any(1).nonzero().flatten()
            print("t", t, nm, "envs", e[:8].tolist(), "cols", neq[e[0]].nonzero().flatten()[:10].tolist(),
                  "a", x[e[0]][neq[e[0]]][:4].tolist(), "b", y[e[0]][neq[e[0]]][:4].tolist(), "done_a", da[e[:8]].tolist())
    if bad:
        break
print("finished at t", t)
