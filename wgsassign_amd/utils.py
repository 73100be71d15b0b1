"""Host-side helpers around the hot path: drop-in for the reference's `utils.py`
(site masks, partition sums, assignment-matrix writer).  Pure NumPy, no device work."""
import gzip

import numpy as np


def preview(names):
    """utils.py:9-14: all names when there are at most four, else the first and last two."""
    names = list(names)
    if len(names) <= 4:
        return ", ".join(names)
    return ", ".join(names[:2]) + ", ..., " + ", ".join(names[-2:])


def print_sample_and_site_summary(sample_names, site_names):
    """utils.py:8-18: first two / last two names of each list."""
    print(f"sample_names: {len(sample_names)} samples total: {preview(sample_names)}")
    print(f"site_names: {len(site_names)} sites total: {preview(site_names)}")


def site_mask(site_names, site_names_target):
    """Boolean keep-mask of utils.py:30-33 (bit-exact: membership of each site in the target).  The reference's
    np.isin sorts both string arrays (6 s at 2M sites); a hash set gives the same booleans in 0.5 s."""
    target = set(site_names_target.tolist() if isinstance(site_names_target, np.ndarray) else site_names_target)
    names = site_names.tolist() if isinstance(site_names, np.ndarray) else site_names
    return np.fromiter((x in target for x in names), dtype=bool, count=len(names))


def filter_sites_to_common(L, site_names, site_names_target):
    """utils.py:22-42: keep the rows of L whose site is in site_names_target."""
    names = np.array(site_names)
    mask = site_mask(names, site_names_target)
    dropped = int(np.sum(~mask))
    if dropped > 0:
        print(f"\tFiltered out {dropped} sites not present in the target site list.")
    return L[mask, :], names[mask].tolist()


def partition_loglikes(per_site_ll, partition_count):
    """utils.py:129-151: sums of a per-site vector by label = site index % partition_count
    (sequential float32 accumulation, like np.add.at)."""
    if per_site_ll.ndim != 1:
        raise ValueError("per_site_ll must be a 1D array")
    labels = np.arange(per_site_ll.shape[0]) % partition_count
    sums = np.zeros(partition_count, dtype=np.float32)
    np.add.at(sums, labels, per_site_ll)
    return sums


def write_ass_mats(filename, loglike_mat, sample_names, pop_names, partition_count=1, print_part_column=True,
                   sample_locations=None, doing_LOO=False):
    """utils.py:49-123: tab-separated assignment matrix, gzipped when the name ends in .gz.  One row per
    (sample, partition) in sample-major order; columns: sample, [source_pop | location], [data_part], then one
    per population.  The reference assembles a DataFrame and calls to_csv(sep="\t", float_format="%.6f"); the
    same bytes are written here row by row through the csv module (which is what to_csv drives): values as
    %.6f, NaN as an empty field, minimal quoting."""
    import csv
    n_ind, K = len(sample_names), len(pop_names)
    values = np.asarray(loglike_mat)
    if values.shape != (n_ind * partition_count, K):
        raise ValueError(f"loglike_mat shape mismatch: expected {(n_ind * partition_count, K)}, got {values.shape}")
    if not print_part_column and partition_count != 1:
        raise ValueError("print_part_column=False is only allowed if partition_count == 1")
    if sample_locations is not None:
        if len(sample_locations) != n_ind:
            raise ValueError("Length of sample_locations does not match sample_names")
        if doing_LOO and not set(sample_locations).issubset(set(pop_names)):
            raise ValueError("sample_locations contains values not in pop_names (required for LOO mode)")
    header = ["sample"]
    if sample_locations is not None:
        header.append("source_pop" if doing_LOO is True else "location")
    if print_part_column:
        header.append("data_part")
    opener = gzip.open if filename.endswith(".gz") else open
    with opener(filename, "wt", newline="") as fh:
        out = csv.writer(fh, delimiter="\t", lineterminator="\n")
        out.writerow(header + [str(p) for p in pop_names])
        for i in range(n_ind):
            lead = [sample_names[i]] + ([sample_locations[i]] if sample_locations is not None else [])
            for part in range(partition_count):
                row = values[i * partition_count + part]
                out.writerow(lead + ([part] if print_part_column else []) + ["" if v != v else "%.6f" % v for v in row])
    print(f"Wrote assignment matrix to {filename}")
