-- An animation script in the vocabulary of the reference's Lua front-end (ch1/src/lua.rs:34-91): a fixed world of random
-- balls over a checked floor, the camera carried once around it, one encoder:AddFrame(world, camera) per step — the
-- one-camera-per-launch sequence rtc_lua_program_render pipelines — and a still with Render at the end.
-- Written for this repository (tests/test_host_cpu.py, tests/test_gpu_round3.py); it needs functions, loops, require and
-- math.random, i.e. the interpreter in csrc/host_lua.cpp.
local orbit = require("orbit_lib")

FRAMES = FRAMES or 12          -- a caller may preset these globals by prepending assignments
BALLS = BALLS or 24
WIDTH, HEIGHT = WIDTH or 320, HEIGHT or 200

local MATT = { ambient = 0.1, diffuse = 0.8, specular = 0.2, shininess = 40.0 }
local MIRROR = { ambient = 0.05, diffuse = 0.4, specular = 0.9, shininess = 250.0, reflectiveness = 0.5 }

world = {
   lights = { { color = { r = 1, g = 1, b = 1 }, position = { x = -6, y = 9, z = -7 } } },
   shapes = {
      { type = "plane", material = { specular = 0, pattern = { type = "checks", color_a = { r = 0.25, g = 0.25, b = 0.25 },
                                                               color_b = { r = 0.75, g = 0.75, b = 0.75 }, scale = 1.5 } } },
      { type = "cube", material = MIRROR, color = { r = 0.8, g = 0.3, b = 0.2 }, rotate_y = 0.6, scale = 0.8, position = { x = 0, y = 0.8, z = 0 } },
   },
}

math.randomseed(13)
for n = 1, BALLS do
   local material = MATT
   if n % 4 == 0 then material = MIRROR end
   table.insert(world.shapes, orbit.random_ball(material, 4.5, 3.0))
end
print(string.format("%d shapes, first ball at (%.4f, %.4f, %.4f)", #world.shapes, world.shapes[3].position.x, world.shapes[3].position.y,
                    world.shapes[3].position.z))

camera = {
   screenwidth = WIDTH, screenheight = HEIGHT,
   position = orbit.on_circle(11, 3.5, 0),
   lookat = { x = 0, y = 1, z = 0 }, up = { x = 0, y = 1, z = 0 },
   fov = math.pi / 3,
}

local film = StartAnimation("orbit.gif")
for frame = 1, FRAMES do
   film:AddFrame(world, camera)
   camera.position = orbit.on_circle(11, 3.5, frame / FRAMES)
end
film:Finish()
print("frames: " .. FRAMES)

camera.position = { x = 0.5, y = 9, z = -4 }   -- a still from high up
Render(world, camera, "orbit_top.ppm")
