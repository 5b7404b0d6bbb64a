"""Soak: the consecutive-layout (preloaded-arguments) step kernel against the general kernel on the same state, at the
full single-GPU sizes, thousands of steps with resets: outputs every 50 steps and the final state bit for bit."""
import os, sys, importlib, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
g = importlib.import_module("gym-acas2d_amd")
same = lambda x, y: torch.equal(x.view(torch.int32), y.view(torch.int32)) if x.is_floating_point() else torch.equal(x, y)
for (E, N, T) in ((65536, 8, 3000), (131072, 8, 600), (65536, 64, 300), (4096, 3, 3000), (1 << 20, 8, 200)):
    gen = torch.Generator(device="cuda:0").manual_seed(3)
    acts = torch.rand(50, E, generator=gen, device="cuda:0") * 2 - 1
    a = g.ACAS2DVecEnv(E, N, device="cuda:0", seed=7, env_offset=11)
    b = g.ACAS2DVecEnv(E, N, device="cuda:0", seed=7, env_offset=11)
    assert a.consecutive_layout, (E, N)
    assert same(a.reset(), b.reset())
    dones = 0
    for t in range(T):
        oa, ra, da, _ = a.step(acts[t % 50])
        os.environ["ACAS2D_NO_ARENA"] = "1"
        ob, rb, db, _ = b.step(acts[t % 50])
        del os.environ["ACAS2D_NO_ARENA"]
        if t % 50 == 0 or t == T - 1:
            assert same(oa, ob) and same(ra, rb) and torch.equal(da, db), (E, N, t)
            for k in ("outcome", "terminal_observation", "episode_return", "episode_steps"):
                assert same(a.outputs[k], b.outputs[k]), (E, N, t, k)
        dones += int(da.sum())
    for name in ("own_x", "own_y", "own_psi", "own_v", "goal_x", "goal_y", "trf_x", "trf_y", "trf_psi", "trf_v", "steps", "total_reward", "episode"):
        assert same(getattr(a, name), getattr(b, name)), (E, N, name)
    print("ok", E, N, T, "episodes finished", dones, "double_buffer", a.double_buffer, flush=True)
