import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lsm_speech_classifier_amd import frontend, synth
from oracle import ref_numpy as O
a = synth.class_chirps(list(range(6)), seed=5); a[3:] = synth.white_noise(3, seed=6)
for F in (40, 128):
    fe = frontend.SpikeFrontEnd(F, "mel")
    p = fe._mel.power(torch.from_numpy(a).cuda()).cpu().numpy()
    db, _ = fe.spectrogram_db(a); r, n = fe.spikes_from_db(db, want_norm=True)
    db, n, r = db.cpu().numpy(), n.cpu().numpy(), r.cpu().numpy()
    e_p = e_db = e_n = 0; flips = 0
    for b in range(6):
        pr = O.mel_power(a[b], F); dr = O.power_to_db(pr); nr = O.normalise_resize(dr)
        e_p = max(e_p, np.abs(p[b] - pr).max() / pr.max()); e_db = max(e_db, np.abs(db[b] - dr).max())
        e_n = max(e_n, np.abs(n[b] - nr).max()); flips += int((r[b] != O.encode_hysteresis(nr, [0.7, 0.8, 0.9, 0.95], 0.1)).sum())
    print(F, "power err/peak %.2e  dB err %.2e  norm err %.2e  raster flips %d of %d" % (e_p, e_db, e_n, flips, r.size))
