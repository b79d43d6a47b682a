"""The oracle's compact LCP (OracleWorld.last_lcp) in the device's layout: one 3-row slot per constraint, frictionless contacts and joint-limit
rows on the slot's first row, upper-limit rows negated."""
import numpy as np


def to_device_layout(L, A, n_contacts):
    m = len(L["b"]); fi = L["findex"]
    dev_of = np.zeros(m, int); sgn = np.ones(m); mu = np.zeros(8)
    slot = 0; r = 0; limMask = negMask = 0; c = 0
    while r < m:
        is_contact = c < n_contacts
        if is_contact and r + 2 < m and fi[r + 1] == r and fi[r + 2] == r:
            dev_of[r:r + 3] = [3 * slot, 3 * slot + 1, 3 * slot + 2]; mu[slot] = L["hi"][r + 1]; r += 3
        else:
            dev_of[r] = 3 * slot
            if not is_contact:
                limMask |= 1 << (3 * slot)
                if np.isinf(L["lo"][r]):
                    negMask |= 1 << (3 * slot); sgn[r] = -1.0
            r += 1
        slot += 1; c += 1
    A24 = np.zeros((24, 24)); b24 = np.zeros(24)
    for i in range(m):
        b24[dev_of[i]] = L["b"][i] * sgn[i]
        for j in range(m):
            A24[dev_of[i], dev_of[j]] = A[i, j] * sgn[i] * sgn[j]
    return {"rows": 3 * slot, "A": A24, "b": b24, "mu": mu, "limMask": limMask, "negMask": negMask, "dev_of": dev_of, "sgn": sgn}
