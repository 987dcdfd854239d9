"""Logger seam of the hot path: `log(msg)` / `logkv(key, val)` as the reference's `cadm.logger.logger` offers them
(/root/reference/cadm/logger/logger.py; the logger itself is out of scope).

The product never imports the reference package.  A caller that runs inside the reference tree hands ITS logger over
once (`set_logger(cadm.logger.logger)`); without one, messages go to stderr and key/values are kept in a
dict (`getkvs()`).
"""
import sys

_kv = {}
_sink = None


def set_logger(logger):
    """Inject an object with `log(msg)` and `logkv(key, val)` (None restores the default)."""
    global _sink
    _sink = logger


def log(msg):
    if _sink is not None:
        _sink.log(msg)
    else:
        print(msg, file=sys.stderr)


def logkv(key, val):
    if _sink is not None:
        _sink.logkv(key, val)
    else:
        _kv[key] = val


def getkvs():
    return dict(_kv)
