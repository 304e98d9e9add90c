"""Minimal reader for the .vtu files csrc/host/vtk_writer.cpp writes (ascii or appended raw), for tests."""
import re
import xml.etree.ElementTree as ET

import numpy as np

DT = {"Float64": np.float64, "Int32": np.int32, "Int64": np.int64, "UInt8": np.uint8}


def read_vtu(path):
    raw = open(path, "rb").read()
    m = re.search(rb'<AppendedData encoding="raw">\s*_', raw)
    appended = None
    if m:
        appended = raw[m.end(): raw.rindex(b"</AppendedData>")]
        raw = raw[: m.start()] + b"</VTKFile>"
    root = ET.fromstring(raw)
    piece = root.find("UnstructuredGrid/Piece")
    out = {"n_points": int(piece.get("NumberOfPoints")), "n_cells": int(piece.get("NumberOfCells")), "arrays": {}}
    for da in piece.iter("DataArray"):
        dt, nc = DT[da.get("type")], int(da.get("NumberOfComponents", "1"))
        if da.get("format") == "ascii":
            a = np.array(da.text.split(), dtype=np.float64).astype(dt)
        else:
            off = int(da.get("offset"))
            nb = int(np.frombuffer(appended[off: off + 8], np.uint64)[0])
            a = np.frombuffer(appended[off + 8: off + 8 + nb], dt).copy()
        out["arrays"][da.get("Name")] = a.reshape(-1, nc) if nc > 1 else a
    return out
