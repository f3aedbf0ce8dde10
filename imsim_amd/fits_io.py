"""Minimal FITS writer / reader for the files downstream of the draw loop: the e-image (float32 primary HDU with
the keywords of imsim/ccd.py:138-204) and the raw amplifier file (primary header of imsim/readout.py:208-299 plus one
int32 IMAGE extension per segment, readout.py:480-533).  astropy is not part of this image; the reference writes the
segments tile-compressed (`CompImageHDU(..., 'RICE_1')`), here they are plain IMAGE extensions with the same pixels
and keywords."""
import numpy as np

BLOCK = 2880


def _fmt_value(v):
    if isinstance(v, bool):
        return ("T" if v else "F").rjust(20)
    if isinstance(v, (int, np.integer)):
        return str(int(v)).rjust(20)
    if isinstance(v, (float, np.floating)):
        s = repr(float(v)).upper()
        if "E" not in s and "." not in s and "N" not in s:
            s += "."
        return s.rjust(20)
    s = str(v).replace("'", "''")
    return "'" + s.ljust(8) + "'"


def card(key, value, comment=None):
    key = str(key).upper()
    if len(key) > 8:
        head = f"HIERARCH {key} = "
    else:
        head = key.ljust(8) + "= "
    text = head + _fmt_value(value)
    if comment:
        text += " / " + comment
    if len(text) > 80:
        text = text[:80]
    return text.ljust(80)


def _header_bytes(cards):
    body = "".join(cards) + "END".ljust(80)
    pad = (-len(body)) % BLOCK
    return (body + " " * pad).encode("ascii")


def _norm_items(header):
    out = []
    for k, v in (header.items() if hasattr(header, "items") else header):
        if isinstance(v, tuple):
            out.append((k, v[0], v[1]))
        else:
            out.append((k, v, None))
    return out


BITPIX = {np.dtype(np.uint8): 8, np.dtype(np.int16): 16, np.dtype(np.int32): 32, np.dtype(np.int64): 64,
          np.dtype(np.float32): -32, np.dtype(np.float64): -64}


def _data_bytes(arr):
    raw = np.ascontiguousarray(arr).astype(arr.dtype.newbyteorder(">"), copy=False).tobytes()
    return raw + b"\0" * ((-len(raw)) % BLOCK)


def hdu_bytes(header, data=None, primary=True):
    """One HDU: mandatory structural keywords first, then the user header (dict or list of pairs; a value may be
    a (value, comment) tuple)."""
    cards = []
    if data is not None:
        data = np.asarray(data)
        bitpix, shape = BITPIX[data.dtype.newbyteorder("=")], data.shape
    else:
        bitpix, shape = 8, ()
    if primary:
        cards.append(card("SIMPLE", True, "conforms to FITS standard"))
    else:
        cards.append(card("XTENSION", "IMAGE", "Image extension"))
    cards.append(card("BITPIX", bitpix))
    cards.append(card("NAXIS", len(shape)))
    for k, n in enumerate(reversed(shape), 1):
        cards.append(card(f"NAXIS{k}", n))
    if primary:
        cards.append(card("EXTEND", True))
    else:
        cards.append(card("PCOUNT", 0))
        cards.append(card("GCOUNT", 1))
    for k, v, c in _norm_items(header):
        cards.append(card(k, v, c))
    out = _header_bytes(cards)
    if data is not None:
        out += _data_bytes(data)
    return out


def write_fits(file_name, hdus):
    """hdus: list of (header, data or None); the first one is the primary HDU."""
    with open(file_name, "wb") as fobj:
        for k, (header, data) in enumerate(hdus):
            fobj.write(hdu_bytes(header, data, primary=(k == 0)))


def _parse_value(text):
    text = text.strip()
    if text.startswith("'"):
        end = 1
        while True:
            end = text.index("'", end)
            if text[end:end + 2] == "''":
                end += 2
                continue
            break
        return text[1:end].replace("''", "'").rstrip()
    val = text.split("/")[0].strip()
    if val in ("T", "F"):
        return val == "T"
    try:
        return int(val)
    except ValueError:
        return float(val)


def read_fits(file_name):
    """-> list of (header dict, data or None)."""
    raw = open(file_name, "rb").read()
    if raw[:2] == b"\x1f\x8b":                     # .fits.gz
        import gzip
        raw = gzip.decompress(raw)
    pos, out = 0, []
    while pos < len(raw):
        header, done = {}, False
        while not done:
            block = raw[pos:pos + BLOCK].decode("ascii")
            pos += BLOCK
            for k in range(0, BLOCK, 80):
                c = block[k:k + 80]
                if c.startswith("END") and c[3:].strip() == "":
                    done = True
                    break
                if c.startswith("HIERARCH"):
                    key, _, rest = c[9:].partition("=")
                    header[key.strip()] = _parse_value(rest)
                elif c[8:10] == "= ":
                    header[c[:8].strip()] = _parse_value(c[10:])
        naxis = header.get("NAXIS", 0)
        data = None
        if header.get("XTENSION") == "BINTABLE":
            # binary table: NAXIS1 x NAXIS2 bytes of rows followed by PCOUNT bytes of heap (variable-length arrays)
            n = int(header["NAXIS1"]) * int(header["NAXIS2"]) + int(header.get("PCOUNT", 0))
            data = _parse_bintable(header, raw[pos:pos + n])
            pos += n + ((-n) % BLOCK)
        elif naxis > 0:
            shape = tuple(header[f"NAXIS{k}"] for k in range(naxis, 0, -1))
            dtype = {v: k for k, v in BITPIX.items()}[header["BITPIX"]]
            n = int(np.prod(shape)) * dtype.itemsize
            data = np.frombuffer(raw[pos:pos + n], dtype=dtype.newbyteorder(">")).astype(dtype).reshape(shape)
            pos += n + ((-n) % BLOCK)
        out.append((header, data))
    return out


# ---- binary tables (cosmic-ray catalogs, imsim/cosmic_rays.py:112-185; tile-compressed images) ----
_TFORM = {"L": ("u1", 1), "B": ("u1", 1), "I": (">i2", 2), "J": (">i4", 4), "K": (">i8", 8), "E": (">f4", 4), "D": (">f8", 8)}


def _split_tform(tform):
    """'1J' -> (1, 'J', None); 'PJ()' / '1PB(120)' -> (1, 'P', 'J')"""
    tform = tform.strip()
    k = 0
    while k < len(tform) and tform[k].isdigit():
        k += 1
    rep = int(tform[:k]) if k else 1
    code = tform[k]
    if code in "PQ":
        return rep, code, tform[k + 1]
    return rep, code, None


def _parse_bintable(header, raw):
    nrow, width = int(header["NAXIS2"]), int(header["NAXIS1"])
    heap0 = int(header.get("THEAP", nrow * width))
    cols, off = {}, 0
    main = np.frombuffer(raw[:nrow * width], dtype=np.uint8).reshape(nrow, width) if nrow else np.zeros((0, width), np.uint8)
    for k in range(1, int(header["TFIELDS"]) + 1):
        rep, code, sub = _split_tform(str(header[f"TFORM{k}"]))
        name = str(header.get(f"TTYPE{k}", f"col{k}"))
        if code in "PQ":
            dsize = 8 if code == "P" else 16
            desc = np.ascontiguousarray(main[:, off:off + dsize]).view(">i4" if code == "P" else ">i8").reshape(nrow, 2)
            dt, isz = _TFORM[sub]
            vals = []
            for n_el, o in desc:
                a = heap0 + int(o)
                vals.append(np.frombuffer(raw[a:a + int(n_el) * isz], dtype=dt).astype(dt.lstrip(">")))
            cols[name] = vals
            off += dsize
        else:
            dt, isz = _TFORM[code]
            block = np.ascontiguousarray(main[:, off:off + rep * isz]).view(dt).reshape(nrow, rep).astype(dt.lstrip(">"))
            cols[name] = block[:, 0] if rep == 1 else block
            off += rep * isz
    return cols


def bintable_hdu_bytes(columns, header=(), extname=None):
    """Binary-table extension from [(name, tform, values)]; tform 'J', 'I', 'E', 'D', 'K', 'B' (scalars per row) or
    'PJ()' / 'PB()' / 'PI()' (one variable-length array per row, stored in the heap)."""
    nrow = len(columns[0][2]) if columns else 0
    parts, heap, tforms = [], bytearray(), []
    for name, tform, values in columns:
        rep, code, sub = _split_tform(tform)
        if code == "P":
            dt, isz = _TFORM[sub]
            desc = np.zeros((nrow, 2), dtype=">i4")
            longest = 0
            for r, v in enumerate(values):
                v = np.asarray(v).astype(dt)
                desc[r] = (len(v), len(heap))
                heap += v.tobytes()
                longest = max(longest, len(v))
            parts.append(desc.view(np.uint8).reshape(nrow, 8))
            tforms.append(f"1P{sub}({longest})")
        else:
            dt, isz = _TFORM[code]
            parts.append(np.ascontiguousarray(np.asarray(values).astype(dt)).view(np.uint8).reshape(nrow, isz))
            tforms.append(f"1{code}")
    main = np.concatenate(parts, axis=1) if parts else np.zeros((0, 0), np.uint8)
    cards = [card("XTENSION", "BINTABLE", "binary table extension"), card("BITPIX", 8), card("NAXIS", 2),
             card("NAXIS1", main.shape[1]), card("NAXIS2", nrow), card("PCOUNT", len(heap)), card("GCOUNT", 1),
             card("TFIELDS", len(columns))]
    for k, ((name, _, _), tf) in enumerate(zip(columns, tforms), 1):
        cards += [card(f"TTYPE{k}", name), card(f"TFORM{k}", tf)]
    if extname:
        cards.append(card("EXTNAME", extname))
    for k, v, c in _norm_items(header):
        cards.append(card(k, v, c))
    body = main.tobytes() + bytes(heap)
    return _header_bytes(cards) + body + b"\0" * ((-len(body)) % BLOCK)
