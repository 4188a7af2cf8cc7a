"""Ordered text output for writers whose target is not a real file (StringIO, wrapped stdout).

Real files go through the library's threaded writers (`_native.text_kmer_matrix`, `text_sig_kmers`, `text_table`:
rows formatted by threads of this process, `csrc/sp_text.hip`).  What is left is formatted here, chunk by chunk, in
this process: a process that owns a HIP context must not spawn copy-on-write children (every later pageable host<->device copy of the
parent paid about a second while such children were alive, and one CLI run in four hung), so there is no
worker pool -- the rows that reach this function are the small or the non-file cases."""


def write_chunks(fout, n_rows, format_rows, chunk=50000):
    """format_rows(lo, hi) -> str for rows [lo, hi); results are written to fout in order."""
    for lo in range(0, max(0, n_rows), chunk):
        fout.write(format_rows(lo, min(lo + chunk, n_rows)))
