"""Stand-ins for bifrost.proclog.ProcLog and bifrost.affinity used by the blocks
(block_base.py:113-119, corr_block.py:336).  ProcLog keeps the latest dict per log name in
memory (PROCLOGS: a reference to the dict the block last passed, so a monitor reads current values) and, when XENG_PROCLOG_DIR is set, also writes bifrost-style `key : value`
files there so a monitor can poll them like /dev/shm/bifrost.  The blocks go on mutating the dict they passed (stats from
two threads in fused mode): a reader takes `snapshot(name)` -- `dict(d)` copies in one step under the interpreter lock -- and
never iterates the live dict."""
import os
import threading

PROCLOGS = {}
_lock = threading.Lock()


class ProcLog:
    def __init__(self, name):
        self.name = name
        self._dir = os.environ.get("XENG_PROCLOG_DIR")        # (read once: update() runs per gulp in every block)
        with _lock:
            PROCLOGS.setdefault(name, {})

    def update(self, contents, *args, **kwargs):
        PROCLOGS[self.name] = contents                        # (the caller's live dict, not a copy: update() runs per gulp in every block)
        d = self._dir
        if d:
            path = os.path.join(d, str(os.getpid()), self.name)
            os.makedirs(os.path.dirname(path), exist_ok=True)
            with open(path, "w") as fh:
                for k, v in dict(contents).items():          # (a copy taken in one step: the block may change its dict meanwhile)
                    fh.write("%s : %s\n" % (k, v))


def snapshot(name):
    """A consistent copy of the latest contents of one log (what a monitor should read)."""
    return dict(PROCLOGS.get(name, {}))


class cpu_affinity:
    @staticmethod
    def set_core(core):
        if core is not None and core >= 0 and hasattr(os, "sched_setaffinity"):
            try:
                os.sched_setaffinity(0, {int(core)})
            except OSError:
                pass

    @staticmethod
    def get_core():
        try:
            cores = sorted(os.sched_getaffinity(0))
            return cores[0] if len(cores) == 1 else -1
        except (AttributeError, OSError):
            return -1
