"""Deterministic scenes for parity tests and bench.py.

Everything is built through the library's [host] constructors, so the structs hold exactly what
the reference's `Sphere::new_with_transform_and_material` etc. would hold.
"""
from __future__ import annotations

import math

from . import Matrix, World, camera, cube, light, material, plane, sphere

MASK64 = (1 << 64) - 1


class SplitMix64:
    """SplitMix64; seed 13 is the seed the reference's own random-scene script uses (ex2.lua:63,69)."""

    def __init__(self, seed: int = 13):
        self.state = seed & MASK64

    def next(self) -> int:
        self.state = (self.state + 0x9E3779B97F4A7C15) & MASK64
        z = self.state
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
        return z ^ (z >> 31)

    def u01(self) -> float:
        return (self.next() >> 11) * (2.0 ** -53)


def synthetic(n_spheres: int, width: int, height: int, with_plane: bool = True, reflective: bool = False,
              seed: int = 13, samples: int = 1):
    """N random spheres (+ checker floor), modelled on the reference's random scene main.rs:321-368.

    Per sphere 9 draws, always consumed: radius, cx, cy, cz, r, g, b, ka, kd.
    `reflective` = config C4: kr 0.3 on every sphere, 0.4 on the plane (depth-5 chains)."""
    rng = SplitMix64(seed)
    w = World(light())
    for _ in range(n_spheres):
        rho = 0.15 + 0.35 * rng.u01()
        cx = -10.0 + 20.0 * rng.u01()
        ucy = rng.u01()
        cy = rho if n_spheres <= 100 else 0.15 + 7.85 * ucy
        cz = -5.0 + 30.0 * rng.u01()
        col = (rng.u01(), rng.u01(), rng.u01())
        ka = 0.05 + 0.25 * rng.u01()
        kd = 0.5 + 0.4 * rng.u01()
        m = material(color=col, ambient=ka, diffuse=kd, specular=0.3, shininess=50.0, reflective=0.3 if reflective else 0.0)
        w.add_shape(sphere(Matrix.identity().scaling(rho, rho, rho).translation(cx, cy, cz), m))
    if with_plane:
        g1, g2 = (0.35, 0.35, 0.35), (0.65, 0.65, 0.65)
        m = material(specular=0.0, reflective=0.4 if reflective else 0.0, pattern=("checker", g1, g2, None))
        w.add_shape(plane(Matrix.identity(), m))
    cam = camera(width, height, 0.7, Matrix.make_view_transform((0.0, 2.0, -8.0), (0.0, 1.0, 5.0), (0.0, 1.0, 0.0)), samples)
    return w, cam


def test7(width: int = 800, height: int = 600):
    """The reference's own `test7` scene (main.rs:204-251): 3 solid spheres + 1 white plane (config C2)."""
    w = World(light())
    w.add_shape(sphere(Matrix.identity().translation(-0.5, 1.0, 0.5),
                       material(color=(1, 0, 0), diffuse=0.7, specular=0.3, shininess=1.0)))
    w.add_shape(sphere(Matrix.identity().scaling(0.5, 0.5, 0.5).translation(1.0, 0.7, -3.5),
                       material(color=(0, 1, 0), diffuse=0.7, specular=0.3, shininess=0.5)))
    w.add_shape(sphere(Matrix.identity().scaling(0.8, 0.8, 0.8).translation(-2.5, 0.53, -0.75).rotation_x(math.pi / 4.0),
                       material(color=(0, 0, 1), diffuse=0.7, specular=0.3, shininess=0.2)))
    w.add_shape(plane(Matrix.identity(), material(color=(1, 1, 1), diffuse=0.7, specular=0.3, shininess=1.0)))
    cam = camera(width, height, math.pi / 2.0, Matrix.make_view_transform((0.0, 0.5, -5.0), (0.0, 1.0, 0.0), (0.0, 1.0, 0.0)))
    return w, cam


def criterion(width: int = 400, height: int = 300):
    """The scene of the reference's Criterion bench (benches/render.rs:10-79): two transparent
    spheres, one opaque, a checker plane; camera looking straight down."""
    w = World(light())
    w.add_shape(sphere(Matrix.identity().translation(-0.5, 1.0, 0.5),
                       material(color=(1, 0, 0), diffuse=0.1, transparency=1.0, refractive_index=1.15, specular=0.1, ambient=0.1)))
    w.add_shape(sphere(Matrix.identity().scaling(0.5, 0.5, 0.5).translation(1.0, 0.7, -3.5),
                       material(color=(0, 1, 0), diffuse=0.1, transparency=1.0, ambient=0.1, refractive_index=1.5, specular=0.1)))
    w.add_shape(sphere(Matrix.identity().scaling(0.8, 0.8, 0.8).translation(-2.5, 0.53, -0.75).rotation_x(math.pi / 4.0),
                       material(color=(0, 0, 1), diffuse=0.7, specular=0.3)))
    w.add_shape(plane(Matrix.identity().translation(0.0, -3.0, 0.0),
                      material(color=(1, 0, 0), diffuse=0.2, ambient=0.6, specular=0.3,
                               pattern=("checker", (1, 1, 1), (0, 0, 0), None))))
    cam = camera(width, height, math.pi / 3.0, Matrix.make_view_transform((0.0, 7.0, 0.0), (0.0, 0.0, 0.0), (1.0, 0.0, 0.0)))
    return w, cam


def test8(width: int = 200, height: int = 150, samples: int = 1):
    """The reference's `test8` scene (main.rs:254-318): three glass spheres (kr and transparency
    both > 0 -> Schlick) over a reflective grid plane."""
    w = World(light())
    w.add_shape(sphere(Matrix.identity().translation(-0.5, 0.0, 0.5),
                       material(color=(1, 0, 0), diffuse=0.1, transparency=1.0, reflective=0.9, refractive_index=1.8,
                                specular=0.9, ambient=0.1)))
    w.add_shape(sphere(Matrix.identity().scaling(0.5, 0.5, 0.5).translation(1.0, 0.7, -1.5),
                       material(color=(0, 1, 0), diffuse=0.1, transparency=1.0, reflective=1.0, ambient=0.1,
                                refractive_index=1.5, specular=0.9, shininess=400.0)))
    w.add_shape(sphere(Matrix.identity().scaling(0.8, 0.8, 0.8).translation(-1.5, 1.0, 1.2),
                       material(color=(0, 0, 1), diffuse=0.07, transparency=1.0, reflective=0.5, specular=0.3,
                                refractive_index=1.25)))
    w.add_shape(plane(Matrix.identity().translation(0.0, -1.5, 0.0),
                      material(color=(1, 0, 0), diffuse=0.2, ambient=0.6, reflective=0.5, specular=0.3,
                               pattern=("grid", (1, 1, 1), (0, 0, 0), Matrix.identity().scaling(2.0, 2.0, 2.0)))))
    cam = camera(width, height, 1.3, Matrix.make_view_transform((0.0, 7.0, -10.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0)), samples)
    return w, cam


def mixed(width: int = 160, height: int = 120, seed: int = 7, n: int = 24):
    """Every shape kind and pattern kind under rotated / sheared / non-uniformly scaled transforms,
    with overlapping glass — exercises cubes, all six patterns, tie-breaking and the n1/n2 open set."""
    rng = SplitMix64(seed)
    u = rng.u01
    w = World(light(position=(-6.0, 9.0, -7.0), intensity=(1.0, 0.95, 0.9)))
    kinds = ["stripe", "gradient", "ring", "checker", "grid", "test"]
    for i in range(n):
        t = (Matrix.identity().scaling(0.3 + u(), 0.3 + u(), 0.3 + u()).rotation_x(3.0 * u()).rotation_y(3.0 * u())
             .rotation_z(3.0 * u()))
        if i % 5 == 0:
            t = t.shearing(0.3 * u(), 0.0, 0.2 * u(), 0.0, 0.0, 0.1 * u())
        t = t.translation(-4.0 + 8.0 * u(), 0.2 + 2.5 * u(), -2.0 + 8.0 * u())
        pat = None
        if i % 3 == 0:
            k = kinds[(i // 3) % len(kinds)]
            pat = (k, (u(), u(), u()), (u(), u(), u()), Matrix.identity().scaling(0.4, 0.4, 0.4).rotation_y(u()))
        glass = (i % 4 == 1)
        m = material(color=(u(), u(), u()), ambient=0.05 + 0.2 * u(), diffuse=0.4 + 0.5 * u(), specular=0.2 + 0.6 * u(),
                     shininess=5.0 + 200.0 * u(), reflective=(0.5 * u() if i % 2 == 0 else 0.0),
                     transparency=(0.4 + 0.5 * u() if glass else 0.0), refractive_index=(1.1 + 0.9 * u() if glass else 1.0),
                     pattern=pat)
        w.add_shape((cube if i % 3 == 2 else sphere)(t, m))
    w.add_shape(plane(Matrix.identity().rotation_y(0.3), material(specular=0.1, reflective=0.25,
                                                                  pattern=("checker", (0.3, 0.3, 0.3), (0.7, 0.7, 0.7), None))))
    w.add_shape(plane(Matrix.identity().rotation_x(math.pi / 2.0).translation(0.0, 0.0, 9.0),
                      material(color=(0.5, 0.6, 0.8), specular=0.0, ambient=0.2)))
    cam = camera(width, height, 1.0, Matrix.make_view_transform((0.5, 2.5, -7.0), (0.0, 1.0, 1.0), (0.0, 1.0, 0.0)))
    return w, cam


def default_scene(width: int = 11, height: int = 11):
    """camera.rs:216-224 test_render1: default world seen from (0,0,-5)."""
    cam = camera(width, height, math.pi / 2.0, Matrix.make_view_transform((0.0, 0.0, -5.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0)))
    return World.default(), cam


def glass_cluster(n: int = 40, width: int = 64, height: int = 48, id_modulus: int = 0, seed: int = 5):
    """Overlapping and nested transparent spheres and cubes (every ray crosses many glass surfaces, so
    compute_refractive's containers walk, shape.rs:115-141, decides most pixels). id_modulus > 0 numbers the
    shapes (index + 1) % id_modulus — shapes then SHARE world ids, the situation the reference's own u8 ids
    produce beyond 255 shapes (shape.rs:287,661-667: 256 -> 0, 257 -> 1, ...)."""
    rng = SplitMix64(seed)
    w = World()
    for i in range(n):
        r = 0.3 + 1.2 * rng.u01()
        t = Matrix.identity().scaling(r, r * (0.6 + 0.8 * rng.u01()), r).translation(-2.5 + 5 * rng.u01(), -1 + 2.5 * rng.u01(), -1 + 4 * rng.u01())
        m = material(color=(rng.u01(), rng.u01(), rng.u01()), diffuse=0.4, ambient=0.1, specular=0.5, shininess=80.0,
                     transparency=0.3 + 0.6 * rng.u01(), reflective=0.3 * rng.u01() if i % 3 else 0.0,
                     refractive_index=1.0 + 0.15 * (i % 7))
        w.add_shape(cube(t, m) if i % 5 == 4 else sphere(t, m))
    w.add_shape(plane(Matrix.identity().translation(0, -2.2, 0), material(color=(0.8, 0.8, 0.8), specular=0.0)))
    if id_modulus > 0:
        for i, sh in enumerate(w.shapes):
            sh.world_id = (i + 1) % id_modulus
    cam = camera(width, height, 0.9, Matrix.make_view_transform((0.0, 0.5, -7.0), (0.0, 0.0, 1.0), (0.0, 1.0, 0.0)))
    return w, cam
