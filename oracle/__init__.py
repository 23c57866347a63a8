"""oracle/ -- TEST INFRASTRUCTURE ONLY (never imported by the product package).

CPU restatement (numpy float64 / torch-CPU float32) of the reference hot path
of henryxrl/Listening-to-Sound-of-Silence-for-Speech-Denoising, used as the
parity checker for the HIP path.  Only `tests/`, `__graft_entry__.smoke()` and
the `cpu_baseline` leg of `bench.py` may import it.

Pinning status
--------------
* Networks, mask ops, bits->mask, add_signals: pinned against the *imported*
  reference modules in the build container (tests/golden/make_goldens.py),
  outputs committed under tests/golden/*.npz.
* STFT/ISTFT arithmetic lives in third-party librosa==0.7.1
  (reference requirements.txt:4) whose source is NOT under /root/reference and
  is not installable here: **parity unpinned** against librosa itself.  The
  restatement follows librosa 0.7.1's published algorithm (core/spectrum.py
  stft/istft, filters.window_sumsquare) anchored on the reference call sites
  (transform.py:174,193,199) and is cross-checked against torch.stft /
  torch.istft in tests/test_oracle_frontend.py.
* Wave decode / mono / resample (oracle/wave_io.py): soundfile + librosa 0.7.1 +
  resampy>=0.2.2, all absent: **parity unpinned**; cross-checked against
  analytic resampling of band-limited tones and scipy.signal.resample_poly in
  tests/test_oracle_wave_io.py.
* Audio-visual variant (nets.video_forward / audiovisual_forward): pinned against
  the reference's live Conv3dBlock / make_video_branch classes evaluated with the
  commented-out configuration (tests/golden/make_goldens_av.py -> audiovisual.npz).
* Objective measures (oracle/metrics.py): pinned against the imported M2/metrics.py
  functions (tests/golden/make_goldens_metrics.py -> metrics.npz); PESQ / STOI not restated.
* Hand-off formats: checked against the reference's own checked-in output files
  (tests/golden/handoff/).
"""
