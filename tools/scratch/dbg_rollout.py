import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import gym_acas2d_amd as g
for (N, E, T) in ((8, 4096, 160), (64, 512, 40)):
    dev = "cuda:0"
    a = g.ACAS2DVecEnv(E, N, device=dev, seed=77, env_offset=5)
    b = g.ACAS2DVecEnv(E, N, device=dev, seed=77, env_offset=5)
    a.reset(); b.reset()
    gen = torch.Generator(device=dev).manual_seed(11)
    actions = torch.rand(T, E, generator=gen, device=dev) * 2 - 1
    out = a.rollout(actions, keep_terminal_obs=True)
    torch.cuda.synchronize()
    for t in range(T):
        obs, rew, done, infos = b.step(actions[t])
        x, y = out["obs"][t], obs
        neq = ~((x == y) | ((x != x) & (y != y)))
        nan = (x != x).sum().item()
        if neq.any() or nan:
            e = neq.any(1).nonzero().flatten()
            print("N", N, "t", t, "nan entries", nan, "mismatching envs", e[:8].tolist(),
                  "cols", neq[e[0]].nonzero().flatten()[:10].tolist() if len(e) else None,
                  "a", x[e[0]][neq[e[0]]][:4].tolist() if len(e) else None, "b", y[e[0]][neq[e[0]]][:4].tolist() if len(e) else None,
                  "done_prev", None)
            if neq.any():
                break
    print("N", N, "finished t", t)
