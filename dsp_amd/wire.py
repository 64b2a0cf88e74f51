"""Result records in the shapes the reference hands to its consumers, so the batched engine can feed the GUI, the
log and the CSV file unchanged.

Reference (SonarWorker.run, airSonar_optimized.py): per frame with at least one valid band it logs
``distance = {dist_kf:.2f} m  (band SNRs: [...], confidence: {avg:.1f}%)`` (:741-743), emits
``distanceSig(dist_kf, list(snrs), avg_confidence)`` (:745), appends the CSV row
``[time.time(), dist_kf, avg_confidence, list(snrs)]`` (:746-747) and, every PLOT_UPDATE_INTERVAL frames, emits the
``waveSig`` dict ``{rx, band_spectra, correlations, band_lengths, rx_id, update_spectrum}`` (:749-757).
``snrs`` holds the SNRs of the bands that produced a range, in band order.

Input everywhere: the structured rows of ``records_to_numpy`` (dtype ``RESULT_DTYPE``).
"""
from typing import Iterable, List, Optional, Sequence

import numpy as np


def _valid_bands(row, n_bands: int) -> List[int]:
    return [j for j in range(n_bands) if (int(row["valid_mask"]) >> j) & 1]


def distance_sig_args(rows: np.ndarray, n_bands: int = 3):
    """Per frame: None when no band produced a range, else ``(dist_kf, list(snrs), avg_confidence)`` (:745)."""
    out = []
    for r in rows:
        bands = _valid_bands(r, n_bands)
        out.append((float(r["kalman"]), [float(r["snr_db"][j]) for j in bands], float(r["avg_conf"])) if bands else None)
    return out


def csv_rows(rows: np.ndarray, timestamps: Optional[Sequence[float]] = None, n_bands: int = 3) -> List[list]:
    """Rows of distances.csv (:746-747): ``[timestamp, dist_kf, avg_confidence, list(snrs)]``, only for frames with a
    valid band.  ``timestamps`` defaults to the frame index."""
    out = []
    for i, sig in enumerate(distance_sig_args(rows, n_bands)):
        if sig is None:
            continue
        dist_kf, snrs, avg = sig
        out.append([float(timestamps[i]) if timestamps is not None else float(i), dist_kf, avg, snrs])
    return out


def csv_text(rows: np.ndarray, timestamps: Optional[Sequence[float]] = None, n_bands: int = 3, header: bool = True) -> str:
    """distances.csv as the reference writes it: the header row of SonarWorker.__init__ (:614-616) and one
    ``csv.writer`` row per frame with a valid band (:746-747)."""
    import csv
    import io
    buf = io.StringIO(newline="")
    w = csv.writer(buf)
    if header:
        w.writerow(["timestamp", "distance", "confidence", "band_snrs"])
    for row in csv_rows(rows, timestamps, n_bands):
        w.writerow(row)
    return buf.getvalue()


def log_lines(rows: np.ndarray, n_bands: int = 3) -> List[Optional[str]]:
    """The INFO line of :741-743 per frame (None when no band produced a range)."""
    out = []
    for sig in distance_sig_args(rows, n_bands):
        if sig is None:
            out.append(None)
            continue
        dist_kf, snrs, avg = sig
        out.append(f"distance = {dist_kf:.2f} m  (band SNRs: {['%.1f' % s for s in snrs]}, confidence: {avg:.1f}%)")
    return out


def wave_sigs(rx: np.ndarray, spec_db: np.ndarray, corr: np.ndarray, first_id: int = 1) -> List[dict]:
    """The waveSig payloads of a batch (:749-757) from ``EchoEngine.chain_wave``'s arrays: rx [B,N], spec_db
    [B,nb,bins], corr [B,nb,L]; ``rx_id`` counts frames from ``first_id`` (update_counter, :748, :755)."""
    return [wave_sig(rx[i], list(spec_db[i]), list(corr[i]), first_id + i) for i in range(len(rx))]


def wave_sig(rx: np.ndarray, band_spectra: Iterable[np.ndarray], correlations: Optional[Iterable[np.ndarray]],
             rx_id: int) -> dict:
    """The waveSig payload of one frame (:749-757).  ``band_spectra``: per band dB spectrum; ``correlations``: per band
    full correlation (or None when the caller did not keep them: zero-length arrays stand in, as the reference
    initialises them, :699)."""
    spectra = [np.asarray(s) for s in band_spectra]
    corrs = [np.asarray(c) for c in correlations] if correlations is not None else [np.array([]) for _ in spectra]
    return {"rx": np.asarray(rx), "band_spectra": spectra, "correlations": corrs,
            "band_lengths": [len(rx)] * len(spectra), "rx_id": int(rx_id), "update_spectrum": True}
