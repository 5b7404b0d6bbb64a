"""CPU tests of the optional renderer (gym-acas2d_amd/render.py; reference ACAS2DGame.view(), game.py:316-431): the
NumPy back-end draws the scene the reference draws, and the pygame back-end is imported only on demand."""
import builtins
import importlib.util

import numpy as np
import pytest

import gym_acas2d_amd as g

R = g.render


def _scene():
    return R.Scene(player=(400.0, 500.0, 0.0, 200.0), traffic=[(900.0, 300.0, 90.0, 200.0), (1200.0, 800.0, 225.0, 210.0)],
                   goal=(1456.0, 500.0), collision_radius=48.0, goal_radius=144.0, steps=17, episode=3, total_reward=12.5,
                   hud={"d_goal": 1056.0, "d_sep": 538.5, "a_lat": -20.0, "v_closing": -3.25, "d_cpa": 100.0,
                        "delta_heading": 1.5, "d_dev": 0.0, "r_step": 0.25, "r_d_goal": 0.1, "r_h_goal": 0.9,
                        "r_d_cpa": 1.0, "r_d_dev": 1.0})


def test_numpy_frame_has_the_reference_layout():
    img = R.rgb_array(_scene())
    assert img.shape == (1000, 1600, 3) and img.dtype == np.uint8               # settings.py:14-15
    assert tuple(img[10, 10]) == R.SKY_RGB                                      # game.py:324
    assert tuple(img[500, 400]) == R.PLAYER_RGB and tuple(img[300, 900]) == R.TRAFFIC_RGB
    assert tuple(img[500, 1456]) == R.GOAL_RGB
    # collision circles of radius 48 around every aircraft, goal circle of radius 144 (game.py:338-346)
    assert tuple(img[500, 400 + 48]) == R.RED_RGB and tuple(img[300 - 48, 900]) == R.RED_RGB
    assert tuple(img[800, 1200 - 48]) == R.RED_RGB and tuple(img[500 - 144, 1456]) == R.YELLOW_RGB
    assert tuple(img[500, 400 + 30]) == R.SKY_RGB                               # the circles are outlines
    # the glyph points along the heading: psi = 0 moves the aircraft by (+x, 0), psi = 90 by (0, +y) (aircraft.py:24-25)
    assert tuple(img[500, 400 + 12]) == R.PLAYER_RGB and tuple(img[500, 400 - 12]) == R.SKY_RGB
    assert tuple(img[300 + 12, 900]) == R.TRAFFIC_RGB and tuple(img[300 - 12, 900]) == R.SKY_RGB


def test_scene_partly_off_screen_is_clipped():
    s = _scene()
    s.traffic.append((-30.0, 1010.0, 10.0, 200.0))
    s.player = (1599.0, 2.0, 300.0, 200.0)
    img = R.rgb_array(s)
    assert img.shape == (1000, 1600, 3) and tuple(img[2, 1598]) == R.PLAYER_RGB


def test_hud_text_follows_the_reference_positions():
    lines = {t.split(":")[0]: (x, y, t) for x, y, t in _scene().text_lines()}
    assert lines["pos"][:2] == (20, 20) and lines["pos"][2] == "pos: (400.0, 500.0)"        # game.py:349-351
    assert lines["Distance to goal"][:2] == (20, 980) and lines["Distance to goal"][2].endswith("1056.0")   # :365-366
    assert lines["Steps"] == (750, 980, "Steps: 17") and lines["Episode"][2] == "Episode: 3"            # :391-394
    assert lines["Step reward"][:2] == (1300, 960) and lines["Step heading reward"][2].endswith("0.900")   # :402,423-427
    assert lines["Total reward"] == (1300, 980, "Total reward: 12.5")


def test_pygame_is_imported_only_on_demand(monkeypatch):
    if importlib.util.find_spec("pygame") is not None:
        pytest.skip("pygame is installed here: the lazy-import guard cannot be exercised")
    real_import = builtins.__import__
    asked = []

    def spy(name, *a, **k):
        if name == "pygame":
            asked.append(name)
        return real_import(name, *a, **k)

    monkeypatch.setattr(builtins, "__import__", spy)
    R.rgb_array(_scene())
    assert not asked                                             # the NumPy path never asks for pygame
    with pytest.raises(R.RendererUnavailable, match="rgb_array"):
        R.PygameWindow()
    assert asked == ["pygame"]


@pytest.mark.gpu
def test_adapter_renders_a_host_copy_of_its_state():
    """ACAS2DEnv.render("rgb_array") (environment.py:50-51 -> game.view()): the frame shows the env's own state, the
    HUD carries the record row the step kernel wrote, and the window back-end asks for pygame only in "human" mode."""
    import random
    import torch
    assert torch.cuda.is_available()
    random.seed(13)
    env = g.ACAS2DEnv(n_traffic=3)
    env.reset()
    for _ in range(5):
        env.step(np.array([0.3]))
    img = env.render(mode="rgb_array")
    p, tr = env.game.player, env.game.traffic
    assert tuple(img[int(round(p.y)), int(round(p.x))]) == R.PLAYER_RGB
    for t in tr:
        if 0 <= t.x < 1600 and 0 <= t.y < 1000:
            assert tuple(img[int(t.y), int(t.x)]) == R.TRAFFIC_RGB
    scene = R.Scene.from_env(env)
    assert scene.steps == env.game.steps == 6 and abs(scene.hud["d_goal"] - env.game.d_goal_record[-1]) < 1e-12
    assert abs(scene.hud["a_lat"] - 0.3 * env.config.acc_lat_limit) < 1e-9
    vec = g.ACAS2DVecEnv(64, 8, device="cuda:0", seed=2)
    vec.reset()
    img = vec.render(index=5)
    assert tuple(img[int(vec.own_y[5].item()), int(vec.own_x[5].item())]) == R.PLAYER_RGB
    if importlib.util.find_spec("pygame") is None:
        with pytest.raises(R.RendererUnavailable):
            env.render(mode="human")
    with pytest.raises(ValueError):
        env.render(mode="ansi")
