-- Helpers for orbit_animation.lua (loaded with require): a module that RETURNS its table.
local lib = {}

lib.TURN = 2 * math.pi

-- Position on a circle of radius `radius` at height `height`, `phase` in [0, 1) of a turn, starting behind the scene.
function lib.on_circle(radius, height, phase)
   local a = phase * lib.TURN
   return { x = radius * math.sin(a), y = height, z = -radius * math.cos(a) }
end

-- A ball of random colour, size and place inside the box [-half, half] x [0.3, top] x [-half, half].
function lib.random_ball(material, half, top)
   local ball = { type = "sphere", material = material }
   ball.color = { r = math.random(), g = math.random(), b = math.random() }
   ball.scale = 0.2 + 0.6 * math.random()
   ball.position = { x = (2 * math.random() - 1) * half, y = 0.3 + (top - 0.3) * math.random(), z = (2 * math.random() - 1) * half }
   return ball
end

return lib
