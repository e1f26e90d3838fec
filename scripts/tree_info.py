import sys, importlib
sys.path.insert(0, '.')
pkg = importlib.import_module('ray-tracer_amd')
sd = pkg.synthetic.stress_scene(10_000_000)
pt = pkg.renderer.HipPathTracer(0); pt.upload_scene(sd)
i = pt.bvh_info()
print('nodes', i.wide_nodes, 'leaves', i.wide_leaves, 'levels', i.wide_levels, 'children/node', (i.wide_nodes - 1 + i.wide_leaves) / i.wide_nodes, 'build s', i.wide_build_seconds)
