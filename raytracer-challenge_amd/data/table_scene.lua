-- A scene in the table vocabulary of the reference's Lua front-end (ch1/src/lua.rs:109-330), written for this repository's
-- tests (tests/test_host_cpu.py::test_lua_table_scene_*): every key the *_from_table functions read is used at least once,
-- keys are deliberately NOT in transform_from_table's application order, and constants use a little arithmetic.
GREY_A = { r = 0.3, g = 0.3, b = 0.3 }
GREY_B = { r = 0.7, g = 0.7, b = 0.7 }
local HALF_TURN = math.pi            --[[ a long comment
   spanning lines ]]
SHINY = { ambient = 0.05, diffuse = 0.6, specular = 0.8, shininess = 120, reflectiveness = 0.25 }

scene = {
   lights = {
      { color = { r = 1, g = 0.9, b = 0.8 }, position = { x = -6, y = 8.5, z = -4 } },
      { color = { r = 1, g = 0, b = 0 }, position = { x = 5, y = 10, z = 10 } },   -- ignored: only lights[1] is read
   },
   shapes = {
      { type = "plane",
        material = { pattern = { type = "checks", color_a = GREY_A, color_b = GREY_B, scale = 0.5, rotate_y = HALF_TURN / 8 },
                     specular = 0, reflectiveness = 0.3 } },
      { type = "sphere", position = { x = -1.2, y = 1, z = 0.4 }, scale = 1, material = SHINY, color = { r = 0.9, g = 0.2, b = 0.2 } };
      { type = "sphere", scale = 0.6, position = { x = 1.1, y = 0.6, z = -0.9 },
        material = { color = { r = 0.05, g = 0.05, b = 0.1 }, ambient = 0, diffuse = 0.3, specular = 0.9, shininess = 300,
                     reflectiveness = 0.8, transparency = 0.85, refractive_index = 1.5 } },
      { type = "cube", position = { x = 2.5, y = 0.5, z = 2 }, rotate_z = 0.1, rotate_x = -0.2, scale = 2 ^ -1, rotate_y = (1 + 2) * 0.25,
        pattern = { type = "stripes", color_a = { r = 0.1, g = 0.6, b = 0.3 }, color_b = { r = 0.9, g = 0.9, b = 0.2 }, scale = 0.2, position = { x = 0.05, y = 0, z = 0 } } },
      { type = "plane", rotate_x = HALF_TURN / 2, position = { x = 0, y = 0, z = 9 }, pattern = { type = "grid", scale = 2 } },
   }
}

view = {
   screenwidth = 96, screenheight = 2 * 32,
   position = { x = -2.5, y = 2.2, z = -6.5 }, lookat = { x = 0, y = 0.8, z = 0 }, up = { x = 0, y = 1, z = 0 },
   fov = HALF_TURN / 3,
   samples = 1,
}

Render(scene, view, "table_scene.ppm")
