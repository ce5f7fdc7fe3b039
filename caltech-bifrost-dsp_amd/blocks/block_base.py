"""Block: the base class of the hot-path blocks, with the reference's public surface
(pipeline/lwa352_pipeline/blocks/block_base.py:95-387): command keys with type/condition
checks, the pending -> active command hand-off under a control lock, JSON command parsing
(etcd callback shape), stats and proclogs.  etcd is optional (`etcd_client=None`), commands
can be injected with process_command_strings() exactly as the reference's tests do
(block_base.py:194-214)."""
import json
import socket
import time
from threading import Lock

from ..proclog import ProcLog, cpu_affinity

COMMAND_OK = 0
COMMAND_NOT_RECOGNIZED = -1
COMMAND_WRONG_TYPE = -2
COMMAND_INVALID = -3


class _Event:
    def __init__(self, value):
        self.value = value


class _WatchResponse:
    def __init__(self, cmds):
        self.events = [_Event(c) for c in cmds]


def declare_streams(ring, *classes):
    """Tell an in-repo ring which of the library's streams touch its spans (ring.py declare_streams; a bifrost ring has no
    such method: nothing to do)."""
    f = getattr(ring, 'declare_streams', None)
    if f is not None:
        f(*classes)


class Block(object):
    pipeline_id = 0
    _instance_count = -1

    @classmethod
    def set_id(cls, x):
        cls.pipeline_id = x

    @classmethod
    def _get_instance_id(cls):
        # per-subclass zero-based counter (block_base.py:80-91)
        cls._instance_count += 1
        return cls._instance_count

    def __init__(self, log, iring, oring, guarantee, core, etcd_client=None,
                 command_keyroot='/cmd/corr', monitor_keyroot='/mon/corr',
                 response_keyroot='/resp/corr', name=None):
        self.log = log
        self.iring, self.oring = iring, oring
        self.guarantee, self.core = guarantee, core
        self.instance_id = self._get_instance_id()
        self.name = name or type(self).__name__
        self.stats = {}
        self.log.info("Pipeline %d: Initializing block: %s (instance %d)" % (self.pipeline_id, self.name, self.instance_id))
        cls = type(self).__name__
        self.bind_proclog = ProcLog(cls + "/bind")
        self.in_proclog = ProcLog(cls + "/in")
        self.out_proclog = ProcLog(cls + "/out")
        self.size_proclog = ProcLog(cls + "/size")
        self.sequence_proclog = ProcLog(cls + "/sequence0")
        self.perf_proclog = ProcLog(cls + "/perf")
        self.stats_proclog = ProcLog(cls + "/stats")
        if self.iring is not None:
            self.in_proclog.update({'nring': 1, 'ring0': self.iring.name})
        if self.oring is not None:
            self.out_proclog.update({'nring': 1, 'ring0': self.oring.name})

        self.etcd_client = etcd_client
        keyfmt = '{root}/x/{host}/pipeline/{pid}/{block}/{id}'
        ids = dict(host=socket.gethostname(), pid=self.pipeline_id, block=self.name, id=self.instance_id)
        self.command_key = keyfmt.format(root=command_keyroot, **ids)
        self.monitor_key = keyfmt.format(root=monitor_keyroot, **ids)
        self.response_key = keyfmt.format(root=response_keyroot, **ids)
        self._etcd_watch_id = None
        self._control_lock = Lock()
        if self.etcd_client:
            self.log.info("Adding watch callback to %s" % self.command_key)
            self._etcd_watch_id = self.etcd_client.add_watch_prefix_callback(self.command_key, self._etcd_callback)
        self.update_pending = False
        self.command_vals = {}
        self._pending_command_vals = {}
        self._command_types = {}
        self._command_conditions = {}
        self._etcd_sets_pending = True

    # ------------------------------------------------------------------ command keys
    def define_command_key(self, name, type=None, condition=None, initial_val=None):
        if initial_val:
            if type:
                assert isinstance(initial_val, type), "%s: key %s: Initial value type check fail!" % (self.name, name)
            if condition:
                assert condition(initial_val), "%s: key %s: Intial value failed condition check! (%s)" % (self.name, name, condition)
        self.command_vals[name] = initial_val
        self._pending_command_vals[name] = initial_val
        self._command_types[name] = type
        self._command_conditions[name] = condition

    def process_command_strings(self, cmds):
        """Process command JSON string(s) as if they had arrived over etcd."""
        if not isinstance(cmds, list):
            cmds = [cmds]
        self._etcd_callback(_WatchResponse(cmds))

    def _parse_event(self, event):
        """-> (seq_id, kwargs) or (seq_id, None) after having sent the error response."""
        v = json.loads(event.value)
        seq_id = v.get('id', None)
        if seq_id is None:
            self._send_command_response("0", False, "Missing ID field")
            return None, None
        if v.get('cmd', None) != "update":
            self._send_command_response("0", False, "Invalid command")
            return None, None
        val = v.get("val", None)
        if not isinstance(val, dict):
            self._send_command_response(seq_id, False, "`val` field should be a dictionary")
            return seq_id, None
        kwargs = val.get("kwargs", None)
        if not isinstance(kwargs, dict):
            self._send_command_response(seq_id, False, "`val[kwargs]` field should be a dictionary")
            return seq_id, None
        return seq_id, kwargs

    def _etcd_callback(self, watchresponse):
        cpu_affinity.set_core(self.core)
        with self._control_lock:
            for event in watchresponse.events:
                seq_id, kwargs = self._parse_event(event)
                if kwargs is None:
                    continue
                try:
                    proc_ok = self._process_commands(kwargs, set_pending_flag=self._etcd_sets_pending)
                except Exception:
                    proc_ok = COMMAND_INVALID
                self.update_stats({'last_cmd_response': proc_ok})
                self._send_command_response(seq_id, proc_ok == COMMAND_OK, str(proc_ok))

    def _send_command_response(self, seq_id, processed_ok, response):
        resp = {'id': seq_id, 'val': {'status': 'normal' if processed_ok else 'error',
                                      'response': response, 'timestamp': time.time()}}
        self.last_response = resp
        if self.etcd_client:
            try:
                self.etcd_client.put(self.response_key, json.dumps(resp))
            except Exception:
                self.log.error("Error trying to send ETCD command response")
                raise
        else:
            self.log.info("No ETCD interface: Command response: %s" % (resp,))

    def _process_commands(self, command_dict, set_pending_flag=True):
        for key, val in command_dict.items():
            if key not in self.command_vals:
                self.log.error("%s: Command key %s not recognized" % (self.name, key))
                return COMMAND_NOT_RECOGNIZED
            if self._command_types[key] and not isinstance(val, self._command_types[key]):
                self.log.error("%s: Command key %s had wrong type (had %s, expected %s)" %
                               (self.name, key, type(val), self._command_types[key]))
                return COMMAND_WRONG_TYPE
            if self._command_conditions[key] and not self._command_conditions[key](val):
                self.log.error("%s: Command key %s failed requirements" % (self.name, key))
                return COMMAND_INVALID
            self._pending_command_vals[key] = val
            self.stats['new_' + key] = val
        if set_pending_flag:
            self.update_pending = True
        self.stats['update_pending'] = True
        self.stats['last_cmd_time'] = time.time()
        return COMMAND_OK

    def update_command_vals(self):
        with self._control_lock:
            self.command_vals.update(self._pending_command_vals)
            self.update_pending = False
            self.stats['update_pending'] = False
            self.stats['last_cmd_proc_time'] = time.time()
        self.update_stats(self.command_vals)

    def acquire_control_lock(self):
        self._control_lock.acquire()

    def release_control_lock(self):
        self._control_lock.release()

    def update_stats(self, new_stats={}):
        self.stats.update(new_stats)
        self.stats_proclog.update(self.stats)
