#!/usr/bin/env python3
"""Hand-assembled OpenFST binary files for the reader tests (tests/test_fst_bytes.py).

Written byte by byte from OpenFST 1.7.5's on-disk layout of VectorFst<StdArc> (the library the
reference links, Makefile:7; fst/fst.h FstHeader::Write, fst/vector-fst.h VectorFst::Write), NOT through
this repo's writer (csrc/fst.cpp, simplefst.py) - the readers have to parse bytes they did not produce:

    int32   magic            2125659606 (kFstMagicNumber)
    string  fst type         int32 length + bytes: "vector"
    string  arc type         int32 length + bytes: "standard"
    int32   version          2 (VectorFst kFileVersion)
    int32   flags            1 = has input symbols, 2 = has output symbols, 4 = aligned (ConstFst only)
    uint64  properties
    int64   start state
    int64   number of states (-1 = unknown: states follow until the stream ends)
    int64   number of arcs
  per state:
    float32 final weight     tropical, +inf = not final
    int64   arc count
    per arc: int32 ilabel, int32 olabel, float32 weight, int32 next state

and, for the ark, Kaldi's table layout of an FST archive: "<key> " then "\\0B" (binary marker) then the FST;
`StdVectorFst.read_ark(filename, offset)` seeks to `offset` and expects the FST magic there
(openfst_binding/src/fstext.cc:7-16), i.e. offset = position after the marker.

The fixtures are data: the .fst / .ark files plus fst_bytes_expected.json (what they contain)."""
import json
import os
import struct

HERE = os.path.dirname(os.path.abspath(__file__))
INF = float("inf")


def header(start, nstates, narcs, flags=0, props=0x0000956A5A950003):
    b = struct.pack("<i", 2125659606)
    b += struct.pack("<i", 6) + b"vector"
    b += struct.pack("<i", 8) + b"standard"
    b += struct.pack("<i", 2)
    b += struct.pack("<i", flags)
    b += struct.pack("<Q", props)
    b += struct.pack("<q", start) + struct.pack("<q", nstates) + struct.pack("<q", narcs)
    return b


def states(spec):
    b = b""
    for final, arcs in spec:
        b += struct.pack("<f", final) + struct.pack("<q", len(arcs))
        for (il, ol, w, ns) in arcs:
            b += struct.pack("<i", il) + struct.pack("<i", ol) + struct.pack("<f", w) + struct.pack("<i", ns)
    return b


# FST A: 4 states, start 1, a self-loop, two arcs into one state, a non-final dead end
A = [(INF, [(3, 3, 0.5, 1), (1, 1, 1.25, 2)]),
     (INF, [(2, 2, 0.25, 0), (5, 5, 2.0, 2), (4, 4, 0.75, 3)]),
     (0.125, [(1, 1, 1.0, 2)]),
     (INF, [])]
# FST B: 3 states, start 0, linear, final weight 1.5
B = [(INF, [(7, 7, 0.1, 1)]), (INF, [(8, 8, 0.2, 1), (9, 9, 0.3, 2)]), (1.5, [])]


def narcs(spec):
    return sum(len(a) for _, a in spec)


def main():
    with open(os.path.join(HERE, "fst_a_plain.fst"), "wb") as f:
        f.write(header(1, len(A), narcs(A)) + states(A))
    # the "aligned" flag is meaningless for a vector FST (no padding is written) but legal in the header;
    # the state count is left unknown (-1), as when OpenFST writes to a stream it cannot seek in
    with open(os.path.join(HERE, "fst_a_flags.fst"), "wb") as f:
        f.write(header(1, -1, 0, flags=4, props=0) + states(A))
    ark = b""
    offsets = {}
    for key, spec, start in (("utt_b", B, 0), ("utt_a", A, 1)):
        ark += key.encode() + b" " + b"\0B"
        offsets[key] = len(ark)
        ark += header(start, len(spec), narcs(spec)) + states(spec)
    with open(os.path.join(HERE, "fst_two.ark"), "wb") as f:
        f.write(ark)
    # symbol tables are not supported: a header that announces one must be rejected, not mis-parsed
    with open(os.path.join(HERE, "fst_with_symbols.fst"), "wb") as f:
        f.write(header(0, len(B), narcs(B), flags=1) + states(B))

    def describe(spec, start):
        return {"start": start, "final": [None if fw == INF else fw for fw, _ in spec],
                "arcs": [[list(a) for a in arcs] for _, arcs in spec]}
    with open(os.path.join(HERE, "fst_bytes_expected.json"), "w") as f:
        json.dump({"fst_a_plain.fst": describe(A, 1), "fst_a_flags.fst": describe(A, 1),
                   "fst_two.ark": {"offsets": offsets, "utt_a": describe(A, 1), "utt_b": describe(B, 0)}}, f, indent=1)


if __name__ == "__main__":
    main()
