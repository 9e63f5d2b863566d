#!/usr/bin/env python3
"""tests/scenes/make_molecules.py -- writes the molecules of our own test page (tests/scenes/page/mol/*.pdb) in the fixed-column
PDB layout Assign07's reader takes (record name 1-6, serial 7-11, atom name 13-16, alt-loc 17, x/y/z 31-54, element 77-78).
Geometry is ours: a double helix of C/N/O/S beads with hydrogens, and a small lattice.  `helix.pdb` also carries the records a
reader has to get right: shuffled serials, a gap in the serials, a duplicate serial, alt-loc 'A' (kept) and 'B' (dropped), a
HETATM, a record too short to have an alt-loc column (dropped), an element with a colour but no radius (I), an element with
neither (XX), a record whose element columns are blank (falls back to the atom name), CONECT / HEADER / TER lines."""
import math
import os

HERE = os.path.dirname(os.path.abspath(__file__))


def rec(kind, serial, name, alt, x, y, z, elem, res="MOL", chain="A", resi=1):
    return f"{kind:<6}{serial:>5} {name:<4}{alt}{res:>3} {chain}{resi:>4}    {x:8.3f}{y:8.3f}{z:8.3f}{1.0:6.2f}{0.0:6.2f}          {elem:>2}"


def helix():
    lines = ["HEADER    SYNTHETIC DOUBLE HELIX", "REMARK   1 written by tests/scenes/make_molecules.py"]
    atoms, serial = [], 1
    for k in range(22):
        a = 0.55 * k
        for strand, el in ((0, "C"), (1, "N")):
            ph = a + math.pi * strand
            atoms.append((serial, el + "A", " ", 4.2 * math.cos(ph), 1.1 * k - 11.0, 4.2 * math.sin(ph), el)); serial += 1
        if k % 3 == 0:
            atoms.append((serial, "O1", " ", 1.4 * math.cos(a + 1.0), 1.1 * k - 11.0, 1.4 * math.sin(a + 1.0), "O")); serial += 1
        if k % 4 == 1:
            atoms.append((serial, "H1", " ", 5.6 * math.cos(a), 1.1 * k - 10.6, 5.6 * math.sin(a), "H")); serial += 1
        if k % 7 == 2:
            atoms.append((serial, "S1", "A", 0.3, 1.1 * k - 11.2, -0.4, "S")); serial += 1           # alt-loc A: kept
            atoms.append((serial, "S2", "B", 0.9, 1.1 * k - 11.2, -0.1, "S")); serial += 1           # alt-loc B: dropped
    serial += 5                                                                                      # a gap in the serials
    atoms.append((serial, "I1", " ", -6.5, 0.0, 2.0, "I")); serial += 1                               # colour, no radius
    atoms.append((serial, "X1", " ", 6.5, 3.0, -2.0, "XX")); serial += 1                              # neither
    atoms.append((serial, "P", " ", -2.0, 9.5, 5.5, "")); serial += 1                                 # blank element -> atom name "P"
    atoms.append((serial, "FE", " ", 2.5, -9.0, -5.0, "FE")); serial += 1                             # colour, no radius
    order = list(range(len(atoms)))
    order = order[1::2] + order[0::2]                                                                # shuffled: file order must not matter
    for i in order:
        s, name, alt, x, y, z, el = atoms[i]
        lines.append(rec("HETATM" if el in ("S", "FE") else "ATOM", s, name, alt, x, y, z, el))
    s0 = atoms[3][0]
    lines.append(rec("ATOM", s0, "CZ", " ", 0.0, 12.5, 0.0, "C"))                                     # duplicate serial: the later record wins
    lines.append("ATOM     77  C")                                                                   # too short for an alt-loc column
    lines.append("TER")
    lines.append(f"CONECT{1:>5}{2:>5}{3:>5}")
    lines.append("END")
    return "\n".join(lines) + "\n"


def lattice():
    lines, serial = ["HEADER    SYNTHETIC LATTICE"], 1
    for i in range(6):
        for j in range(6):
            for k in range(6):
                el = ("C", "O", "N", "H")[(i + 2 * j + 3 * k) % 4]
                lines.append(rec("ATOM", serial, el, " ", 2.9 * i + 0.3 * ((j + k) % 2), 2.9 * j + 0.2 * (k % 3), 2.9 * k - 0.25 * (i % 2), el, resi=1 + serial // 10))
                serial += 1
    return "\n".join(lines) + "\nEND\n"


def main():
    out = os.path.join(HERE, "page", "mol")
    os.makedirs(out, exist_ok=True)
    open(os.path.join(out, "helix.pdb"), "w").write(helix())
    open(os.path.join(out, "lattice.pdb"), "w").write(lattice())


if __name__ == "__main__":
    main()
