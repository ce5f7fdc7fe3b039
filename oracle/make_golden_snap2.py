#!/usr/bin/env python3
"""Generate the committed SNAP2 packet fixture under tests/golden/.

TEST INFRASTRUCTURE ONLY.  Pins oracle.snap2_packets / snap2_unpack (and, on the GPU, xengSnap2Unpack) to bytes
that the REFERENCE's own F-engine emulator emits.  Runs in the build container only (/root/reference does not
exist on the GPU box):

  1. the reference's golden generator (pipeline/verification/make_golden_inputs.py, --nocorr) writes an input
     file of 8 spectra x 4 channels x 64 stands x 2 pols (seed 0xdeadbeef);
  2. the reference's packet transmitter (pipeline/test_transmitters/test_tx_vectors.py) is run on that file with
     `socket.socket` replaced by a recorder: every datagram it would send is kept, and after one pass over the
     file (ntime x nchan_blocks x npol_blocks packets) the recorder raises KeyboardInterrupt, which is how the
     script's endless send loop is meant to be stopped (test_tx_vectors.py:95-121).

Outputs (data only -- no reference source is stored):
  tests/golden/in_8t_4c_64s_2p_deadbeef.dat     the generator's input file (JSON header line + uint8[t][c][s][p])
  tests/golden/snap2_8t_4c_64s_2p_deadbeef.bin  JSON header line + the recorded datagrams, concatenated
"""
import json
import os
import runpy
import socket
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import GOLD, run_ref, strip_time  # noqa: E402

TX = "/root/reference/pipeline/test_transmitters/test_tx_vectors.py"


class Recorder:
    """Stands in for socket.socket(AF_INET, SOCK_DGRAM) inside the reference script."""
    packets = []
    limit = 0

    def __init__(self, *a, **k):
        pass

    def sendto(self, data, addr):
        if len(Recorder.packets) >= Recorder.limit:
            raise KeyboardInterrupt
        Recorder.packets.append(bytes(data))


def main():
    nt, nc, ns, npol, nchan_blocks = 8, 4, 64, 2, 2
    with tempfile.TemporaryDirectory() as td:
        run_ref(["-t", str(nt), "-c", str(nc), "-s", str(ns), "-p", str(npol), "--accshort", str(nt), "--nocorr"], td)
        fin = os.path.join(td, "in_%dt_%dc_%ds_%dp_deadbeef.dat" % (nt, nc, ns, npol))
        strip_time(fin)
        Recorder.packets = []
        Recorder.limit = nt * nchan_blocks * (ns // 32)
        real_socket, argv = socket.socket, sys.argv
        socket.socket = Recorder
        sys.argv = [TX, "-f", fin, "-b", str(nchan_blocks)]
        try:
            runpy.run_path(TX, run_name="__main__")
        finally:
            socket.socket, sys.argv = real_socket, argv
        os.replace(fin, os.path.join(GOLD, os.path.basename(fin)))
    pk = Recorder.packets
    assert len(pk) == Recorder.limit and len(set(map(len, pk))) == 1
    meta = {"npkt": len(pk), "pkt_bytes": len(pk[0]), "ntime": nt, "nchan": nc, "nstand": ns, "npol": npol,
            "nchan_blocks": nchan_blocks, "nstand_per_pkt": 32,
            "source": "test_tx_vectors.py run on in_%dt_%dc_%ds_%dp_deadbeef.dat under a recording socket" % (nt, nc, ns, npol)}
    out = os.path.join(GOLD, "snap2_%dt_%dc_%ds_%dp_deadbeef.bin" % (nt, nc, ns, npol))
    with open(out, "wb") as fh:
        fh.write(json.dumps(meta).encode() + b"\n")
        fh.write(b"".join(pk))
    print(out, os.path.getsize(out), meta)


if __name__ == "__main__":
    main()
