"""Key-value logger with the call shape of baselines/logger.py:191-219 (logkv / dumpkvs),
writing a plain table to stdout.  The other writers of the reference are out of scope."""
import sys
from collections import OrderedDict

_kvs = OrderedDict()
_dir = None
_quiet = False


def configure(dir=None, quiet=False):
    global _dir, _quiet
    _dir, _quiet = dir, quiet


def get_dir():
    return _dir


def logkv(key, val):
    _kvs[key] = val


def logkvs(d):
    for k, v in d.items():
        logkv(k, v)


def getkvs():
    return _kvs


def dumpkvs(out=None):
    if not _quiet and _kvs:
        out = out or sys.stdout
        w = max(len(k) for k in _kvs)
        out.write("-" * (w + 20) + "\n")
        for k, v in _kvs.items():
            s = "%.6g" % v if isinstance(v, float) else str(v)
            out.write("| %-*s | %-14s |\n" % (w, k, s))
        out.write("-" * (w + 20) + "\n")
    _kvs.clear()


def log(*args):
    if not _quiet:
        print(*args)
