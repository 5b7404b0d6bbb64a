"""Optional CPU renderer, split out of the GPU path.

The reference draws every frame inside the game loop with pygame (``ACAS2DGame.view()``,
gym_ACAS2D/envs/game.py:316-431: sky, the three sprites, collision / goal circles, the player's state and
the metrics / sub-rewards as text).  Here rendering never touches the step path: the caller copies ONE
env's state to the host (`Scene.from_env`) and hands it to a back-end --

* ``rgb_array(scene)``  a NumPy rasteriser (no third-party window system): the picture as uint8 [H, W, 3];
* ``PygameWindow``      a window like the reference's, imported lazily: without pygame installed (it is not
  in this image) constructing it raises `RendererUnavailable` with that explanation instead of an ImportError
  at package import.

Aircraft are drawn as heading-oriented triangles (the reference's PNG sprites are not part of this build);
the window size, colours, circle radii and text layout follow settings.py:14-25 and game.py:323-428.
"""
from dataclasses import dataclass, field

import numpy as np

WIDTH, HEIGHT = 1600, 1000                         # settings.py:14-15
SKY_RGB, BLACK_RGB, RED_RGB, YELLOW_RGB = (60, 150, 220), (0, 0, 0), (255, 0, 0), (255, 255, 0)   # settings.py:20-24
PLAYER_RGB, TRAFFIC_RGB, GOAL_RGB = (255, 255, 255), (40, 40, 40), (0, 255, 0)
AIRCRAFT_SIZE = 24                                  # settings.py:33


class RendererUnavailable(RuntimeError):
    pass


@dataclass
class Scene:
    """Host copy of one env: everything `ACAS2DGame.view()` reads (game.py:316-431)."""
    player: tuple                 # (x, y, psi, v_air)
    traffic: list                 # [(x, y, psi, v_air), ...]
    goal: tuple                   # (x, y)
    collision_radius: float       # COLLISION_RADIUS = collision_dist / 2 (game.py:187, settings.py:34)
    goal_radius: float
    steps: int = 0
    episode: int = 0
    total_reward: float = 0.0
    hud: dict = field(default_factory=dict)         # name -> value: the metrics / sub-rewards text of game.py:349-425
    acc_lat_limit: float = 196.133                  # ACC_LAT_LIMIT = 20 g (settings.py:42): the a_lat_norm line
    dt: float = 0.01                                # 1 / FPS (aircraft.py:18): the psi_dot line

    @classmethod
    def from_env(cls, env, index=0):
        """From an `ACAS2DEnv` (its record row supplies the HUD numbers) or an `ACAS2DVecEnv` (env `index`)."""
        vec = getattr(env, "_vec", env)
        i = int(index)
        host = lambda t: float(t[i].item())  # noqa: E731
        cfg = vec.config
        traffic = [tuple(float(v) for v in row) for row in zip(*(getattr(vec, n)[i].cpu().tolist()
                                                                 for n in ("trf_x", "trf_y", "trf_psi", "trf_v")))]
        hud = {}
        if getattr(vec, "trace", None) is not None:
            from .vec_env import TRACE_COLUMNS
            hud = dict(zip(TRACE_COLUMNS, vec.trace[i].cpu().tolist()))
        game = getattr(env, "game", None)
        return cls(player=(host(vec.own_x), host(vec.own_y), host(vec.own_psi), host(vec.own_v)), traffic=traffic,
                   goal=(host(vec.goal_x), host(vec.goal_y)), collision_radius=float(cfg.collision_radius),
                   goal_radius=float(cfg.goal_radius), steps=int(vec.steps[i].item()),
                   episode=int(getattr(game, "episode", None) or vec.episode[i].item()),
                   total_reward=host(vec.total_reward), hud=hud, acc_lat_limit=float(cfg.acc_lat_limit),
                   dt=float(cfg.dt))

    def text_lines(self):
        """The HUD as (x, y, text) in the reference's positions (game.py:349-404)."""
        px, py, psi, v = self.player
        h = self.hud
        left = [(20, 20, "pos: (%.1f, %.1f)" % (px, py)), (20, 40, "v_air: %.1f" % v), (20, 60, "psi: %.1f" % psi)]
        if "a_lat" in h:                             # game.py:355-361: psi_dot = a_lat / (v_air dt) (aircraft.py:20)
            left += [(20, 80, "psi_dot: %.1f" % (h["a_lat"] / (v * self.dt))), (20, 100, "a_lat: %.1f" % h["a_lat"]),
                     (20, 120, "a_lat_norm: %.3f" % (h["a_lat"] / self.acc_lat_limit))]
        bottom = [("d_goal", "Distance to goal", 20), ("d_sep", "Min. Separation", 40), ("v_closing", "Closing Speed", 80),
                  ("d_cpa", "Closest approach", 100), ("delta_heading", "Delta heading", 120), ("d_dev", "Plan deviation", 140)]
        left += [(20, HEIGHT - dy, "%s: %.1f" % (label, h[k])) for k, label, dy in bottom if k in h]
        if self.traffic:                             # game.py:369-372: relative_angle(player -> traffic[0]) (kinematics.py:16-22)
            tx, ty = self.traffic[0][:2]
            left.append((20, HEIGHT - 60, "Rel. angle to traffic: %.1f" % (np.degrees(np.arctan2(ty - py, tx - px) % (2 * np.pi)))))
        mid = [(WIDTH // 2 - 50, HEIGHT - 20, "Steps: %d" % self.steps), (WIDTH // 2 - 50, HEIGHT - 40, "Episode: %s" % self.episode)]
        right = [(WIDTH - 300, HEIGHT - 20, "Total reward: %.1f" % self.total_reward)]
        rewards = [("r_step", "Step reward", 40), ("r_d_dev", "Step plan deviation reward", 60),
                   ("r_d_goal", "Step goal distance reward", 80), ("r_d_cpa", "Step closest approach reward", 100),
                   ("r_h_goal", "Step heading reward", 120)]
        right += [(WIDTH - 300, HEIGHT - dy, "%s: %.3f" % (label, h[k])) for k, label, dy in rewards if k in h]
        return left + mid + right


def _triangle(x, y, psi_deg, size=AIRCRAFT_SIZE):
    """Vertices of an aircraft glyph at (x, y) pointing along its heading (screen y grows downwards, and the
    engine moves an aircraft by (cos psi, sin psi) in those coordinates: aircraft.py:24-25)."""
    a = np.deg2rad(psi_deg)
    c, s = np.cos(a), np.sin(a)
    nose, left, right = (0.6 * size, 0.0), (-0.4 * size, -0.35 * size), (-0.4 * size, 0.35 * size)
    return [(x + u * c - w * s, y + u * s + w * c) for u, w in (nose, left, right)]


# ---- NumPy back-end ---------------------------------------------------------------------------------
def _fill_polygon(img, pts, rgb):
    pts = np.asarray(pts, np.float64)
    x0, x1 = int(max(np.floor(pts[:, 0].min()), 0)), int(min(np.ceil(pts[:, 0].max()) + 1, img.shape[1]))
    y0, y1 = int(max(np.floor(pts[:, 1].min()), 0)), int(min(np.ceil(pts[:, 1].max()) + 1, img.shape[0]))
    if x0 >= x1 or y0 >= y1:
        return
    yy, xx = np.mgrid[y0:y1, x0:x1]
    inside = np.ones(yy.shape, bool)
    sign = None
    for k in range(len(pts)):
        (ax, ay), (bx, by) = pts[k], pts[(k + 1) % len(pts)]
        cross = (bx - ax) * (yy + 0.5 - ay) - (by - ay) * (xx + 0.5 - ax)
        if sign is None:
            sign = 1.0 if (pts[(k + 2) % len(pts)][1] - ay) * (bx - ax) - (pts[(k + 2) % len(pts)][0] - ax) * (by - ay) >= 0 else -1.0
        inside &= cross * sign >= 0
    img[y0:y1, x0:x1][inside] = rgb


def _ring(img, cx, cy, r, rgb, width=1.0):
    x0, x1 = int(max(np.floor(cx - r - 1), 0)), int(min(np.ceil(cx + r + 2), img.shape[1]))
    y0, y1 = int(max(np.floor(cy - r - 1), 0)), int(min(np.ceil(cy + r + 2), img.shape[0]))
    if x0 >= x1 or y0 >= y1:
        return
    yy, xx = np.mgrid[y0:y1, x0:x1]
    d = np.hypot(xx + 0.5 - cx, yy + 0.5 - cy)
    img[y0:y1, x0:x1][np.abs(d - r) <= width / 2 + 0.5] = rgb


def rgb_array(scene, width=WIDTH, height=HEIGHT):
    """The frame of game.py:323-345 (no text) as uint8 [height, width, 3]."""
    img = np.empty((height, width, 3), np.uint8)
    img[:] = SKY_RGB
    gx, gy = scene.goal
    half = AIRCRAFT_SIZE / 2
    _fill_polygon(img, [(gx - half, gy - half), (gx + half, gy - half), (gx + half, gy + half), (gx - half, gy + half)], GOAL_RGB)
    for (x, y, psi, _v) in scene.traffic:
        _fill_polygon(img, _triangle(x, y, psi), TRAFFIC_RGB)
    px, py, ppsi, _ = scene.player
    _fill_polygon(img, _triangle(px, py, ppsi), PLAYER_RGB)
    _ring(img, px, py, scene.collision_radius, RED_RGB)                       # game.py:338
    _ring(img, gx, gy, scene.goal_radius, YELLOW_RGB)                         # game.py:341-342
    for (x, y, _psi, _v) in scene.traffic:
        _ring(img, x, y, scene.collision_radius, RED_RGB)                     # game.py:345-346
    return img


# ---- pygame back-end --------------------------------------------------------------------------------
class PygameWindow:
    """The reference's window (game.py:42-55, :316-431).  `draw(scene)` returns False once the user closed it
    (the `quit` flag the reference's scripts poll, testing_main.py:66)."""

    def __init__(self, caption="ACAS-2D", font_size=14):
        try:
            import pygame  # noqa: PLC0415
        except ImportError as e:
            raise RendererUnavailable(
                "render(mode='human') needs pygame, which is not installed; the step engine does not depend on it -- "
                "use render(mode='rgb_array') (NumPy) or install pygame") from e
        self._pg = pygame
        pygame.init()
        self.screen = pygame.display.set_mode((WIDTH, HEIGHT))
        pygame.display.set_caption(caption)                                    # settings.py:17
        self.font = pygame.font.Font(None, font_size + 4)
        self.open = True

    def draw(self, scene):
        pg = self._pg
        for event in pg.event.get():
            if event.type == pg.QUIT:
                self.open = False
        # the reference's order (game.py:323-346): sky, player, goal, traffic, then the player's collision circle, the
        # goal circle, the traffic's collision circles; glyphs stand in for its three 24 x 24 sprites
        self.screen.fill(SKY_RGB)
        gx, gy = scene.goal
        px, py, ppsi, _ = scene.player
        pg.draw.polygon(self.screen, PLAYER_RGB, _triangle(px, py, ppsi))
        pg.draw.rect(self.screen, GOAL_RGB, (gx - AIRCRAFT_SIZE / 2, gy - AIRCRAFT_SIZE / 2, AIRCRAFT_SIZE, AIRCRAFT_SIZE))
        for (x, y, psi, _v) in scene.traffic:
            pg.draw.polygon(self.screen, TRAFFIC_RGB, _triangle(x, y, psi))
        pg.draw.circle(self.screen, RED_RGB, (px, py), scene.collision_radius, 1)
        pg.draw.circle(self.screen, YELLOW_RGB, (gx, gy), scene.goal_radius, 1)
        for (x, y, _psi, _v) in scene.traffic:
            pg.draw.circle(self.screen, RED_RGB, (x, y), scene.collision_radius, 1)
        for x, y, text in scene.text_lines():
            self.screen.blit(self.font.render(text, True, BLACK_RGB), (x, y))
        pg.display.update()
        return self.open

    def close(self):
        if self.open:
            self._pg.display.quit()
        self.open = False
