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
        if naxis > 0:
            shape = tuple(header[f"NAXIS{k}"] for k in range(naxis, 0, -1))
            dtype = {v: k for k, v in BITPIX.items()}[header["BITPIX"]]
            n = int(np.prod(shape)) * dtype.itemsize
            data = np.frombuffer(raw[pos:pos + n], dtype=dtype.newbyteorder(">")).astype(dtype).reshape(shape)
            pos += n + ((-n) % BLOCK)
        out.append((header, data))
    return out
