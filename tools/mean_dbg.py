import importlib, sys, os, time
sys.path.insert(0, os.getcwd())
pkg = importlib.import_module('2048_amd')
B = 1 << 20
eng = pkg.Engine(B, n=5, seed=2048)
eng.init_weights(seed=7, scale=0.01)
eng.td_steps(0.25 * 21 / (8.0 * B), 300)
eng.set_update_rule(1)
eng.td_steps(0.25, 64)
eng.sync()
for rep in range(3):
    t0 = time.perf_counter()
    eng.timer_start()
    eng.td_steps(0.25, 200)
    ms = eng.timer_stop()
    dt = time.perf_counter() - t0
    print(f'200 steps: events {ms / 200:.4f} ms/step, wall {dt / 200 * 1e3:.4f} ms/step', flush=True)
print('kernel ms (play, owner, tail, apply):', eng.td_steps_kernel_ms(0.25, 20))
plan = eng.debug_owner_plan()
print('workgroups', len(plan), 'chunks', len(set(plan[:,1].tolist())), 'min nparts', plan[:,3].min(), 'max', plan[:,3].max())
