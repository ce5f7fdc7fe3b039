"""Integration-boundary bookkeeping shared by Corr and CorrAcc.

Both reference blocks carry the same hand-rolled state machine (corr_block.py:350-466,
corr_acc_block.py:211-332) with two quirks each; here it is one small class with the quirks
as parameters:

                         Corr                                   CorrAcc
  step                   ntime_gulp                             upstream acc_len
  start_time == -1       next multiple of acc_len after `now`   `now` itself
                         (corr_block.py:397-398)                (corr_acc_block.py:244-245)
  recovery after a new   last_start + (missed + 10) * acc_len   last_start + (missed + 2) * acc_len
  upstream sequence      (corr_block.py:360-366)                (corr_acc_block.py:221-227)
"""


class IntegrationGate:
    def __init__(self, recovery_skip, round_start_to_acc_len):
        self.recovery_skip = recovery_skip
        self.round_start = round_start_to_acc_len
        self.running = False
        self.start_time = 0
        self.acc_len = 0
        self.first = self.last = None

    def recover(self, seq0):
        """A new upstream sequence arrived while integrating: realign to a boundary in the future.
        Returns True if a recovery start time was set."""
        if not self.running:
            return False
        if self.acc_len > 0:
            missed = (seq0 - self.start_time) // self.acc_len
            self.start_time += (missed + self.recovery_skip) * self.acc_len
        self.running = False
        return True

    def configure(self, now, acc_len, start_cmd):
        """New command values (corr_block.py:392-404): sets acc_len / start_time, stops integrating."""
        self.acc_len = acc_len
        if start_cmd == -1:
            if self.round_start and acc_len > 0:
                self.start_time = now - (now % acc_len) + acc_len
            else:
                self.start_time = now
        else:
            self.start_time = start_cmd
        self.running = False

    def try_start(self, now, step):
        """True when `now` is the commanded start: begins integrating, sets first/last."""
        if now != self.start_time:
            return False
        self.running = True
        self.first = self.start_time
        self.last = self.first + self.acc_len - step
        return True

    def advance(self, step):
        """After the last block of an integration: the next one follows immediately."""
        self.first = self.last + step
        self.last = self.first + self.acc_len - step
