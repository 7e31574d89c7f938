import importlib, os, sys
sys.path.insert(0, '/root/repo')
pkg = importlib.import_module('2048_amd')
B = 1 << 20
eng = pkg.Engine(B, n=5, seed=2048)
eng.init_weights(seed=7, scale=0.01)
alpha = 0.25 * 21 / (8.0 * B)
eng.td_steps(alpha, 320)
print('normal     ', ' '.join(f'{x:.4f}' for x in eng.td_steps_kernel_ms(alpha, 32)))
print('alpha = 0  ', ' '.join(f'{x:.4f}' for x in eng.td_steps_kernel_ms(0.0, 32)))
print('normal     ', ' '.join(f'{x:.4f}' for x in eng.td_steps_kernel_ms(alpha, 32)))
