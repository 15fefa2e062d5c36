"""Logger with the reference's format and env switches (/root/reference/src/classpose/log.py:5-53)."""
import logging
import os
import sys

FMT = "%(asctime)s,%(msecs)03d %(name)s %(levelname)s %(message)s"


def get_logger(name: str) -> logging.Logger:
    logger = logging.getLogger(name)
    if getattr(logger, "_cpx_configured", False):
        return logger
    rank = int(os.environ.get("RANK", 0))
    level = os.environ.get("LOG_LEVEL", "INFO") if rank == 0 else os.environ.get("LOG_LEVEL_NON_MAIN", "WARNING")
    logger.setLevel(getattr(logging, level.upper(), logging.INFO))
    h = logging.StreamHandler(sys.stderr)
    h.setFormatter(logging.Formatter(FMT, datefmt="%Y-%m-%d %H:%M:%S"))
    logger.addHandler(h)
    path = os.environ.get("CLASSPOSE_LOG_PATH")
    if path:
        fh = logging.FileHandler(path)
        fh.setFormatter(logging.Formatter(FMT, datefmt="%Y-%m-%d %H:%M:%S"))
        logger.addHandler(fh)
    logger.propagate = False
    logger._cpx_configured = True
    return logger


def apply_rank_level() -> None:
    """Re-apply the rank-dependent level to every logger configured by ``get_logger`` (ranks forked from a parent inherit its loggers,
    configured before ``RANK`` was set: non-zero ranks log at ``LOG_LEVEL_NON_MAIN`` like the reference's workers, log.py:5-53)."""
    rank = int(os.environ.get("RANK", 0))
    level = os.environ.get("LOG_LEVEL", "INFO") if rank == 0 else os.environ.get("LOG_LEVEL_NON_MAIN", "WARNING")
    for lg in list(logging.Logger.manager.loggerDict.values()):
        if isinstance(lg, logging.Logger) and getattr(lg, "_cpx_configured", False):
            lg.setLevel(getattr(logging, level.upper(), logging.INFO))
