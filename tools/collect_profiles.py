#!/usr/bin/env python3
"""Copy what tools/r04_profiles.sh assembled on the GPU box (gpurun_out/r04_profiles/profiles_out/) into the tracked
profiles/ directory — the step round 2 did "by hand".

    python tools/collect_profiles.py [gpurun_out/r04_profiles/profiles_out]
"""
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'gpurun_out', 'r04_profiles', 'profiles_out')
dst = os.path.join(ROOT, 'profiles')
for name in sorted(os.listdir(src)):
    shutil.copy(os.path.join(src, name), os.path.join(dst, name))
    print('profiles/' + name, os.path.getsize(os.path.join(dst, name)), 'bytes')
