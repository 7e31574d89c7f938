"""Platform glue of the reference (game2048/start.py), reduced to what the hot path's callers use.

The reference's start.py (a) star-exports its imports to game_logic / r_learning / show.py, (b) owns the
UI-thread registries GAME_PANE / AGENT_PANE / RUNNING, (c) talks to an S3 bucket through boto3.  (a) and (b) are
kept as they are part of the surface; (c) — cloud I/O, out of scope — is replaced by the
same function names over a local directory ($G2048_STORAGE, default ./storage), so `show.py`-style drivers
that list / load / save agents and games keep working on a box with no network.
"""
from datetime import datetime, timedelta  # noqa: F401  (re-exported like the reference does)
import json
import os
import pickle
import random  # noqa: F401
import sys  # noqa: F401
import time  # noqa: F401
from collections import deque  # noqa: F401
from pprint import pprint  # noqa: F401
from threading import Thread  # noqa: F401

import numpy as np  # noqa: F401

working_directory = os.path.dirname(os.path.realpath(__file__))
# game2048/config.json:19-26 (UI timer intervals); only check_thread() in r_learning needs them
CONF = {'intervals': {'refresh_sec': 5, 'vc_sec': 300}}
LOCAL = os.environ.get('S3_URL', 'local')
dash_intervals = CONF['intervals']
dash_intervals['refresh'] = dash_intervals['refresh_sec'] * 1000          # start.py:24-27
dash_intervals['check_run'] = dash_intervals['refresh_sec'] * 2
dash_intervals['vc'] = dash_intervals['vc_sec'] * 1000
dash_intervals['next'] = dash_intervals['refresh_sec'] + 180
LOWEST_SPEED = 50

GAME_PANE = {}          # start.py:30-32
AGENT_PANE = {}
RUNNING = {}

STORAGE = os.environ.get('G2048_STORAGE', os.path.join(os.getcwd(), 'storage'))


def _path(name):
    return os.path.join(STORAGE, name)


def time_suffix(precision=1):                      # start.py:54-55
    return ''.join([v for v in str(datetime.utcnow()) if v.isnumeric()])[4:-precision]


def next_time():                                   # start.py:58-59
    return str(datetime.utcnow() + timedelta(seconds=dash_intervals['next']))


def list_names_s3():                               # start.py:67-68
    out = []
    for root, _, files in os.walk(STORAGE):
        for f in files:
            out.append(os.path.relpath(os.path.join(root, f), STORAGE).replace(os.sep, '/'))
    return sorted(out)


def is_data_there(name):                           # start.py:71-72
    return os.path.exists(_path(name))


def delete_s3(name):                               # start.py:79-81
    if is_data_there(name):
        os.remove(_path(name))


def load_s3(name):                                 # start.py:84-101: json / txt / pkl by extension, None if absent
    if not is_data_there(name):
        return None
    ext = name.rsplit('.', 1)[-1]
    if ext == 'json':
        with open(_path(name), 'r', encoding='utf-8') as f:
            return json.load(f)
    if ext == 'txt':
        with open(_path(name), 'r') as f:
            return f.read()
    with open(_path(name), 'rb') as f:
        return pickle.load(f)


def save_s3(data, name):                           # start.py:104-119
    os.makedirs(os.path.dirname(_path(name)) or STORAGE, exist_ok=True)
    ext = name.rsplit('.', 1)[-1]
    if ext == 'json':
        with open(_path(name), 'w', encoding='utf-8') as f:
            json.dump(data, f, ensure_ascii=False, indent=4)
    elif ext == 'txt':
        with open(_path(name), 'w') as f:
            f.write(data)
    elif ext == 'pkl':
        with open(_path(name), 'wb') as f:
            pickle.dump(data, f, -1)
    else:
        return 0
    return 1


class Logger:                                      # start.py:144-158: an appendable text object
    msg = {'welcome': 'Welcome! Let\'s do something interesting. Choose MODE of action!',
           'stop': 'Process terminated by user'}

    def __init__(self, log_file='logs.txt', start=''):
        self.file = log_file
        save_s3(start, self.file)

    def add(self, text):
        if text:
            save_s3((load_s3(self.file) or '') + '\n' + str(text), self.file)
