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


# ---- RICE_1 tile-compressed images (the reference writes the 16 segments of a raw file with
#      fits.CompImageHDU(..., compression_type='RICE_1'), imsim/readout.py:500-510) ----
# FITS tiled-image convention: a binary table with one row per tile (here: one image row per tile, the default) whose
# COMPRESSED_DATA column holds the Rice-coded bytes; keywords ZIMAGE, ZCMPTYPE, ZBITPIX, ZNAXISn, ZTILEn, ZNAMEn / ZVALn
# (BLOCKSIZE 32, BYTEPIX 4).  The Rice coder is the 32-bit variant of CFITSIO's fits_rcomp: the first pixel verbatim in
# 32 bits, then per block of 32 pixels the zig-zag mapped differences to the previous pixel, split at FS bits: FS + 1 in
# 5 bits, then for every pixel the high part in unary (that many zeros and a one) and the low FS bits verbatim; FS = 0
# with an all-zero block is coded as 0 alone, FS >= 25 switches to 32 raw bits per difference.
RICE_BLOCK = 32


def _rice_fs(diff_blocks, counts):
    """split position per block from the block sums (fits_rcomp: dpsum = (sum - n/2 - 1) / n, FS = bit length of dpsum / 2)"""
    psum = diff_blocks.sum(axis=1, dtype=np.float64)
    dpsum = (psum - (counts // 2) - 1) / counts
    dpsum = np.where(dpsum < 0, 0.0, dpsum)
    half = (dpsum.astype(np.uint64) >> np.uint64(1)).astype(np.uint64)
    fs = np.zeros(len(half), dtype=np.int64)
    nz = half > 0
    fs[nz] = np.floor(np.log2(half[nz].astype(np.float64))).astype(np.int64) + 1
    # guard the float log2 at exact powers of two
    fs = np.where((half >> fs.astype(np.uint64)) > 0, fs + 1, fs)
    fs = np.where((fs > 0) & ((half >> (fs - 1).clip(0).astype(np.uint64)) == 0), fs - 1, fs)
    return fs, psum


def rice_compress_rows(img):
    """Rice-code every row of an int32 image as its own tile.  Returns a list of uint8 arrays (one per row)."""
    img = np.ascontiguousarray(img, dtype=np.int32)
    ny, nx = img.shape
    a = img.astype(np.int64)
    prev = np.concatenate([a[:, :1], a[:, :-1]], axis=1)
    pd = (a - prev).astype(np.int32).astype(np.int64)                   # 32-bit wrap-around differences, as in C
    diff = np.where(pd < 0, ~(pd << 1), pd << 1) & 0xFFFFFFFF           # zig-zag to unsigned
    nb = (nx + RICE_BLOCK - 1) // RICE_BLOCK
    pad = nb * RICE_BLOCK - nx
    d = np.pad(diff, ((0, 0), (0, pad))).reshape(ny * nb, RICE_BLOCK)
    counts = np.full(ny * nb, RICE_BLOCK, dtype=np.int64)
    if pad:
        counts[nb - 1::nb] = RICE_BLOCK - pad
    valid = np.arange(RICE_BLOCK)[None, :] < counts[:, None]
    fs, psum = _rice_fs(d, counts)
    raw = fs >= 25
    zero = (fs == 0) & (psum == 0)
    # bit runs per pixel: `top` zeros, a one, then the low fs bits; per block a 5-bit header.  All rows are laid out in
    # ONE bit array (every row padded to a whole byte) and filled group by group of equal FS.
    fsb = np.where(raw | zero, 0, fs)
    top = np.where(valid, d >> fsb[:, None], 0)
    coded = ~(raw | zero)
    ln = np.where(raw[:, None], 32, np.where(zero[:, None], 0, top + 1 + fsb[:, None])) * valid          # bits per pixel
    head = np.where(raw, 26, np.where(zero, 0, fs + 1))
    per_block = np.concatenate([np.full((ny * nb, 1), 5, dtype=np.int64), ln], axis=1)                # header + pixels
    per_row = per_block.reshape(ny, nb * (RICE_BLOCK + 1))
    starts_in_row = 32 + np.cumsum(per_row, axis=1) - per_row
    row_bits = 32 + per_row.sum(axis=1)
    row_bytes = (row_bits + 7) // 8
    row_base = np.concatenate([[0], np.cumsum(row_bytes * 8)[:-1]])
    starts = (starts_in_row + row_base[:, None]).reshape(ny * nb, RICE_BLOCK + 1)
    bits = np.zeros(int(row_bytes.sum()) * 8, dtype=np.uint8)
    # first pixel of every row, 32 bits
    first = (img[:, 0].astype(np.int64) & 0xFFFFFFFF)
    bits[(row_base[:, None] + np.arange(32)[None, :]).reshape(-1)] = ((first[:, None] >> np.arange(31, -1, -1)[None, :]) & 1).reshape(-1)
    # block headers, 5 bits
    bits[(starts[:, :1] + np.arange(5)[None, :]).reshape(-1)] = ((head[:, None] >> np.arange(4, -1, -1)[None, :]) & 1).reshape(-1)
    pix_start = starts[:, 1:]
    sel = coded[:, None] & valid
    bits[(pix_start + top)[sel]] = 1                                         # the one that ends the unary part
    for f in np.unique(fsb[coded]):
        f = int(f)
        if f == 0:
            continue
        m = (coded & (fsb == f))[:, None] & valid
        base = (pix_start + top + 1)[m]
        vals = d[m]
        bits[(base[:, None] + np.arange(f)[None, :]).reshape(-1)] = ((vals[:, None] >> np.arange(f - 1, -1, -1)[None, :]) & 1).reshape(-1)
    if raw.any():
        m = raw[:, None] & valid
        base = pix_start[m]
        vals = d[m]
        bits[(base[:, None] + np.arange(32)[None, :]).reshape(-1)] = ((vals[:, None] >> np.arange(31, -1, -1)[None, :]) & 1).reshape(-1)
    packed = np.packbits(bits)
    offs = np.concatenate([[0], np.cumsum(row_bytes)])
    return [packed[offs[r]:offs[r + 1]] for r in range(ny)]


def rice_decompress_row(buf, nx):
    """inverse of the coder above for one tile (used by the tests and by read_compressed_image)"""
    bits = np.unpackbits(np.asarray(buf, dtype=np.uint8))
    pos = 0

    def take(n):
        nonlocal pos
        v = 0
        for b in bits[pos:pos + n]:
            v = (v << 1) | int(b)
        pos += n
        return v
    last = take(32)
    if last >= 1 << 31:
        last -= 1 << 32
    out = np.zeros(nx, dtype=np.int64)
    i = 0
    while i < nx:
        n = min(RICE_BLOCK, nx - i)
        fs = take(5) - 1
        for j in range(n):
            if fs < 0:
                diff = 0
            elif fs == 25:
                diff = take(32)
            else:
                t = 0
                while bits[pos] == 0:
                    t += 1
                    pos += 1
                pos += 1
                diff = (t << fs) | (take(fs) if fs else 0)
            pd = (diff >> 1) if (diff & 1) == 0 else ~(diff >> 1)
            last = ((last + pd + (1 << 31)) % (1 << 32)) - (1 << 31)
            out[i + j] = last
        i += n
    return out.astype(np.int32)


def compressed_image_hdu_bytes(header, data, extname=None):
    """int32 image -> tile-compressed BINTABLE extension (ZCMPTYPE RICE_1, one row per tile)"""
    data = np.ascontiguousarray(data, dtype=np.int32)
    ny, nx = data.shape
    rows = rice_compress_rows(data)
    z = [("ZIMAGE", (True, "extension contains compressed image")), ("ZTENSION", "IMAGE"), ("ZBITPIX", 32), ("ZNAXIS", 2),
         ("ZNAXIS1", nx), ("ZNAXIS2", ny), ("ZPCOUNT", 0), ("ZGCOUNT", 1), ("ZTILE1", nx), ("ZTILE2", 1), ("ZCMPTYPE", "RICE_1"),
         ("ZNAME1", "BLOCKSIZE"), ("ZVAL1", RICE_BLOCK), ("ZNAME2", "BYTEPIX"), ("ZVAL2", 4)]
    user = [(k, (v, c) if c else v) for k, v, c in _norm_items(header) if k != "EXTNAME"]
    name = extname or dict((k, v) for k, v, _ in _norm_items(header)).get("EXTNAME")
    return bintable_hdu_bytes([("COMPRESSED_DATA", "PB()", rows)], header=z + user, extname=name)


def write_fits_compressed(file_name, hdus):
    """like write_fits, the image extensions RICE_1 tile-compressed (the primary HDU stays as it is)"""
    with open(file_name, "wb") as fobj:
        for k, (header, data) in enumerate(hdus):
            if k == 0 or data is None:
                fobj.write(hdu_bytes(header, data, primary=(k == 0)))
            else:
                fobj.write(compressed_image_hdu_bytes(header, data))


def read_compressed_image(header, table):
    """(header, parsed BINTABLE) of a RICE_1 tile-compressed int32 image -> the image"""
    nx, ny = int(header["ZNAXIS1"]), int(header["ZNAXIS2"])
    if header.get("ZCMPTYPE") != "RICE_1" or int(header.get("ZTILE2", 1)) != 1 or int(header["ZTILE1"]) != nx:
        raise ValueError("only row-tiled RICE_1 images are supported")
    return np.stack([rice_decompress_row(row, nx) for row in table["COMPRESSED_DATA"]]).reshape(ny, nx)
