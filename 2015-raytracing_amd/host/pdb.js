// host/pdb.js -- the molecule reader behind Assign07's sphere mode (SURVEY 8f rank 4).
// Reference behaviour restated: A07 mol/pdbParserV1.js:2-85 (what a record contributes, which lookups miss, what `size` means);
// own code: one pass over fixed-column records into a serial-indexed table, then one pass in key order.
//
// What the packers downstream rely on, quirks included:
//   * only ATOM / HETATM records with an alternate-location column of ' ' or 'A' count; a record shorter than 17 columns has an
//     EMPTY alt-loc column and is therefore dropped (pdbParserV1.js:20-21)
//   * atoms are keyed by serial-1, a later record with the same serial replaces the earlier one, and the result is walked in key
//     order, so file order does not matter (:37, :59)
//   * `size` is the largest key + 1, not the atom count (:78): hosts that loop to `size` read past `atomData` (undefined -> NaN),
//     which the grid builder's floor()/compare logic turns into "no cell"
//   * an element without a van-der-Waals entry has radius `undefined` (NaN once packed): it takes part in nothing;
//     an element without a colour entry is black (hex2rgb(undefined) = 0,0,0)
//   * element = columns 77-78 without blanks, else the atom-name columns 13-16 without blanks (:31-34)
"use strict";
const { Bounds } = require("./scene.js");

// CPK-like colours and Bondi radii (J. Phys. Chem. 68 (1964) 441), keyed exactly as the page keys them (case matters)
const ELEMENT_RGB = { H: 0xCCCCCC, C: 0xAAAAAA, O: 0xCC0000, N: 0x0000CC, S: 0xCCCC00, P: 0x6622CC, F: 0x00CC00, CL: 0x00CC00, BR: 0x882200,
                      I: 0x6600AA, FE: 0xCC6600, CA: 0x8888AA };
const VDW_RADIUS = { H: 1.2, Li: 1.82, Na: 2.27, K: 2.75, C: 1.7, N: 1.55, O: 1.52, F: 1.47, P: 1.80, S: 1.80, CL: 1.75, BR: 1.85, SE: 1.90,
                     ZN: 1.39, CU: 1.4, NI: 1.63 };

const noBlanks = (s) => s.split(" ").join("");

function parsePDB(text) {
  const table = [];   // sparse, index = serial - 1
  for (const raw of text.split("\n")) {
    const rec = raw.replace(/^\s*/, "");
    const tag = rec.substr(0, 6);
    if (tag !== "ATOM  " && tag !== "HETATM") continue;   // CONECT bonds and HEADER text are not used by any packer
    const alt = rec.substr(16, 1);
    if (alt !== " " && alt !== "A") continue;
    let elem = noBlanks(rec.substr(76, 2));
    if (elem === "") elem = noBlanks(rec.substr(12, 4));
    table[parseInt(rec.substr(6, 5)) - 1] = { elem: elem, x: parseFloat(rec.substr(30, 8)), y: parseFloat(rec.substr(38, 8)), z: parseFloat(rec.substr(46, 8)) };
  }
  const typeOf = {}, colorData = [], radiusData = [], atomData = [];
  const lo = [Number.MAX_VALUE, Number.MAX_VALUE, Number.MAX_VALUE], hi = [-Number.MAX_VALUE, -Number.MAX_VALUE, -Number.MAX_VALUE];
  for (const key in table) {   // integer keys ascending, then any non-index key (a NaN serial) in insertion order
    const a = table[key];
    if (typeOf[a.elem] === undefined) {
      const hex = ELEMENT_RGB[a.elem];
      colorData.push(((hex >> 16) & 255) / 255, ((hex >> 8) & 255) / 255, (hex & 255) / 255, 1);
      typeOf[a.elem] = radiusData.length;
      radiusData.push(VDW_RADIUS[a.elem]);
    }
    const t = typeOf[a.elem], R = radiusData[t], p = [a.x, a.y, a.z];
    atomData.push(t, a.x, a.y, a.z);
    for (let c = 0; c < 3; c++) {
      if (p[c] - R < lo[c]) lo[c] = p[c] - R;
      if (p[c] + R > hi[c]) hi[c] = p[c] + R;
    }
  }
  return { size: table.length, atomData: atomData, colorData: colorData, radiusData: radiusData, bounds: new Bounds(lo, hi) };
}

module.exports = { parsePDB, ELEMENT_RGB, VDW_RADIUS };
