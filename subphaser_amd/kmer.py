"""k-mer <-> uint64 codec (2 bits/base, A=0 C=1 G=2 T=3, first base most significant)."""
import numpy as np

_LUT = np.full(256, 255, np.uint8)
for _i, _ch in enumerate("ACGT"):
    _LUT[ord(_ch)] = _i
    _LUT[ord(_ch.lower())] = _i
_CHARS = np.frombuffer(b"ACGT", np.uint8)


def encode(kmer):
    v = 0
    for ch in kmer:
        c = _LUT[ord(ch)]
        if c == 255:
            raise ValueError("non-ACGT character in k-mer %r" % kmer)
        v = (v << 2) | int(c)
    return v


def encode_many(kmers):
    """list of equal-length str -> uint64 array"""
    if len(kmers) == 0:
        return np.empty(0, np.uint64)
    k = len(kmers[0])
    a = np.frombuffer("".join(kmers).encode(), np.uint8).reshape(len(kmers), k)
    codes = _LUT[a]
    if (codes == 255).any():
        raise ValueError("non-ACGT character in k-mer list")
    out = np.zeros(len(kmers), np.uint64)
    for j in range(k):
        out = (out << np.uint64(2)) | codes[:, j].astype(np.uint64)
    return out


def decode(key, k):
    key = int(key)
    return "".join("ACGT"[(key >> (2 * (k - 1 - i))) & 3] for i in range(k))


def decode_many(keys, k):
    """uint64 array -> list of str"""
    keys = np.asarray(keys, np.uint64)
    if keys.size == 0:
        return []
    out = np.empty((keys.size, k), np.uint8)
    for j in range(k):
        out[:, j] = _CHARS[((keys >> np.uint64(2 * (k - 1 - j))) & np.uint64(3)).astype(np.intp)]
    return out.view("S%d" % k).ravel().astype(str).tolist()


def revcomp(keys, k):
    """reverse complement of uint64-coded k-mers (vectorised)"""
    x = np.asarray(keys, np.uint64)
    out = np.zeros_like(x)
    three = np.uint64(3)
    for _ in range(k):
        out = (out << np.uint64(2)) | (three - (x & three))
        x = x >> np.uint64(2)
    return out


def canonical(keys, k):
    keys = np.asarray(keys, np.uint64)
    return np.minimum(keys, revcomp(keys, k))


def dense_slots(k):
    """size of the dense count table (csrc/sp_common.h sp_dense_slots)"""
    return 1 << (2 * k - 1) if k % 2 else 1 << (2 * k)


def slots_of_keys(keys, k):
    """dense-table slot of k-mers (any orientation); mirrors sp_slot_of_key in csrc/sp_common.h:
    odd k  -> the strand whose middle base is A/C, with the high bit of that base removed;
    even k -> the canonical value itself."""
    keys = np.asarray(keys, np.uint64)
    rc = revcomp(keys, k)
    if k % 2 == 0:
        return np.minimum(keys, rc)
    one = np.uint64(1)
    rep = np.where(((keys >> np.uint64(k)) & one).astype(bool), rc, keys)
    low = rep & ((one << np.uint64(k)) - one)
    return low | ((rep >> np.uint64(k + 1)) << np.uint64(k))


def keys_of_slots(slots, k):
    """canonical k-mer of dense-table slots (sp_key_of_slot)"""
    slots = np.asarray(slots, np.uint64)
    if k % 2 == 0:
        return slots.copy()
    one = np.uint64(1)
    rep = (slots & ((one << np.uint64(k)) - one)) | ((slots >> np.uint64(k)) << np.uint64(k + 1))
    return np.minimum(rep, revcomp(rep, k))
