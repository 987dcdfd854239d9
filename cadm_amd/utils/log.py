"""Minimal stand-in for the reference's logger calls made from the hot path
(/root/reference/cadm/logger/logger.py `log` / `logkv`; the logger itself is out of scope,
SURVEY.md section 2 row 10).  If the reference package is importable its logger is used."""
import sys

_kv = {}

try:  # drop-in inside the reference tree
    from cadm.logger import logger as _ref_logger  # type: ignore
except Exception:  # pragma: no cover - the reference is not importable in this image
    _ref_logger = None


def log(msg):
    if _ref_logger is not None:
        _ref_logger.log(msg)
    else:
        print(msg, file=sys.stderr)


def logkv(key, val):
    if _ref_logger is not None:
        _ref_logger.logkv(key, val)
    else:
        _kv[key] = val


def getkvs():
    return dict(_kv)
