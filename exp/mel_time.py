import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lsm_speech_classifier_amd import frontend, synth
import bench
B = 256
audio = torch.from_numpy(bench.make_audio("speech_like", B, 1234)).cuda()
for F in (40, 128):
    fe = frontend.SpikeFrontEnd(F, "mel")
    for _ in range(2): fe.encode(audio)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    ev[0].record(); p = fe._mel.power(audio); ev[1].record(); db = fe._mel.power_db(audio); ev[2].record(); r, _ = fe.spikes_from_db(db); ev[3].record()
    torch.cuda.synchronize()
    print(f"mel F={F} B={B}: stft+mel {ev[0].elapsed_time(ev[1]):.3f} ms, (stft+mel again)+power_to_db {ev[1].elapsed_time(ev[2]):.3f} ms, spec_to_spikes {ev[2].elapsed_time(ev[3]):.3f} ms -> {B/((ev[1].elapsed_time(ev[2])+ev[2].elapsed_time(ev[3]))*1e-3):.0f} clips/s front end")
