"""Minimal stat logger with the reference's `log_stat` / `print_recent_stats` surface (src/utils/logging.py:5-58).
tensorboard / sacred sinks are out of scope."""
import logging
from collections import defaultdict


class Logger:
    def __init__(self, console_logger=None):
        self.console_logger = console_logger or get_logger()
        self.stats = defaultdict(list)

    def log_stat(self, key, value, t, to_sacred=True):
        self.stats[key].append((t, value))

    def print_recent_stats(self):
        if "episode" not in self.stats:
            return
        t, ep = self.stats["episode"][-1]
        items = ["t_env: {:>10} | Episode: {:>8}".format(t, ep)]
        for k in sorted(self.stats):
            if k == "episode":
                continue
            window = [float(v) for _, v in self.stats[k][-5:]]
            items.append("{}: {:.4f}".format(k, sum(window) / len(window)))
        self.console_logger.info("Recent Stats | " + " | ".join(items))


def get_logger():
    logger = logging.getLogger("ssd-hip")
    if not logger.handlers:
        h = logging.StreamHandler()
        h.setFormatter(logging.Formatter("[%(levelname)s %(asctime)s] %(name)s %(message)s", "%H:%M:%S"))
        logger.addHandler(h)
        logger.setLevel(logging.INFO)
    return logger
