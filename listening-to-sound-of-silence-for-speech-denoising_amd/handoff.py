"""The on-disk hand-off between the two models, so the build interoperates stage by stage
with the reference's scripts on the README's real-recording path (`--unknown_clean_signal true`):

  dataset JSON (PP/tools.py:28-31; data/sounds_of_silence.json)
    -> detect_files                 M1/predict.py:38-233  `evaluate(clean_audio=False)`   -> eval_results.json
    -> create_data_from_prediction  M1/create_data_from_pred.py:38-271 (clean_audio=False) -> pred_data.json + recovered/*_mixed.wav
    -> get_data_from_first_model    M2/predict.py:255-374 (unknown_clean_signal=True)
    -> denoise_files                M2/predict.py:377-576                                  -> <id>/{noisy_input,noise_intervals,
                                                                     predicted_full_noise,denoised_output}.wav, stat.json, eval_results.json

Same keys, key order, value types and JSON formatting as the reference writes (checked against the reference's
own checked-in outputs in tests/golden/handoff/).  Both branches of the scripts are covered: real recordings
(`--unknown_clean_signal true`) and clean recordings mixed with noise at an SNR (clean_audio=True: noise bookkeeping,
`_mixed / _clean / _full_noise` WAVE files, objective measures in stat.json).  The PNG plots (cv2 / matplotlib) are
not built; the `waveform` / `spectrum` keys are therefore absent from stat.json.  All signal work (decode, resample,
STFT, networks, masks, ISTFT, measures) runs on the GPU through audio_io / transform / tools / metrics; this module is
the host-side bookkeeping around it."""
import json
import os
from collections import OrderedDict
from itertools import groupby
from operator import itemgetter

import numpy as np
import torch

from . import audio_io, metrics, tools, transform
from .tools import add_signals

JSON_DUMP_PARAMS = dict(indent=4, sort_keys=False, ensure_ascii=False, separators=(',', ':'))   # M1/tools.py:36
BITSTREAM_JSON_LABEL = 'bit_stream'            # M1/tools.py:54
BIT_STREAM_LABEL = 'recovered_prediction'      # M2/predict.py:30
GT_BIT_STREAM_LABEL = 'bit_stream'             # M2/predict.py:32
CLIP_FRAMES = 60                               # M1/dataset.py:33
SILENT_CONSECUTIVE_FRAMES = 1                  # M1/dataset.py:32
SIGMOID_THRESHOLD = 0.5                        # M1/predict.py:30
DATA_REQUIRED_SR = 14000                       # M1/dataset.py:38


def ensure_dir(path):
    os.makedirs(path, exist_ok=True)


def get_parent_dir(path):
    return os.path.abspath(os.path.join(path, os.pardir))


def find_common_path(str1, str2, sep='/'):
    """M1/utils.py:189-191."""
    return os.path.commonprefix([str1, str2]).rpartition(sep)[0]


def convert_snr_to_suffix2(snr):
    """M1/tools.py:882-891: None -> '', 10.0 -> '_snr10', 2.5 -> '_snr2_5'."""
    if snr is None:
        return ""
    try:
        snr = float(snr)
    except (TypeError, ValueError):
        return ""
    return '_snr' + str(int(snr) if snr.is_integer() else snr).replace('.', '_')


def convert_threshold_to_suffix(threshold_str):
    """M1/create_data_from_pred.py:26-35."""
    try:
        threshold = float(threshold_str)
    except (TypeError, ValueError):
        return ""
    return '_' + str(threshold).replace('.', '_') if 0 <= threshold <= 1 else ""


def show_metrics(y_true, y_score):
    """M1/tools.py:91-197.  Labels: 1 = non-silent, 0 = silent; SILENT is the positive class of the
    confusion counts.  Rates whose denominator is zero come out as NaN -> JSON null (the reference divides
    numpy scalars, which yields nan rather than raising)."""
    y_true = np.asarray(y_true).astype(np.int64)
    y_score = np.asarray(y_score).astype(np.int64)
    n = len(y_true)
    n_silent = int(np.sum(y_true == 0))
    n_non_silent = int(np.sum(y_true == 1))
    base = n_non_silent / n
    accuracy = int(np.sum(y_true == y_score)) / n
    t, s = 1 - y_true, 1 - y_score
    tp = int(np.sum(t * s))
    fp = int(np.sum((t == 0) * s))
    tn = int(np.sum((t == 0) * (s == 0)))
    fn = int(np.sum(t * (s == 0)))

    def div(a, b):
        return float(a) / float(b) if b else float('nan')

    tpr = div(tp, tp + fn)
    fpr = div(fp, fp + tn)
    precision = div(tp, tp + fp)
    tnr = 1 - fpr
    f1 = div(2 * tp, 2 * tp + fp + fn)
    auc = (tpr + tnr) / 2
    den = float(np.sqrt(float(tp + fp) * float(tp + fn) * float(tn + fp) * float(tn + fn)))
    mcc = 0.0 if den == 0 else (tp * tn - fp * fn) / den

    def nan_to_null(v):
        return None if isinstance(v, float) and np.isnan(v) else v

    return OrderedDict([
        ('num_samples', n), ('num_silent_samples', n_silent), ('num_non_silent_samples', n_non_silent),
        ('base', base), ('accuracy', accuracy),
        ('true_positive', tp), ('false_positive', fp), ('true_negative', tn), ('false_negative', fn),
        ('true_pos_rate(recall)', nan_to_null(tpr)), ('false_pos_rate', nan_to_null(fpr)),
        ('precision', nan_to_null(precision)), ('true_neg_rate', nan_to_null(tnr)), ('f1', nan_to_null(f1)),
        ('roc_auc', nan_to_null(auc)), ('mcc', nan_to_null(float(mcc)))])


_trim_unknown = tools.trim_unknown_frames


def _resolve(path, dataset_path, data_root):
    """The JSONs carry the authors' absolute paths; `data_root` re-roots them onto this machine."""
    if data_root is None or not dataset_path or not path.startswith(dataset_path):
        return path
    return os.path.join(data_root, os.path.relpath(path, dataset_path))


# ------------------------------------------------------------------------------------------- model 1 -> JSON
@torch.no_grad()
def add_noise_to_audio(audio, noise, snr, start_pos=0, norm=0.5):
    """M1/tools.py:846-869 with an explicit start position: crop (zero-pad) the noise to the audio, mix at `snr` dB,
    peak-normalise to `norm` (add_signals, M1/tools.py = M2/tools.py:217-276).  numpy in, numpy out."""
    crop = np.asarray(noise)[start_pos:start_pos + len(audio)]
    if len(crop) < len(audio):
        crop = np.concatenate((crop, np.zeros(len(audio) - len(crop), dtype=crop.dtype)))
    return add_signals(np.asarray(audio), [crop], snr=snr, norm=norm)


def detect_files(net, dataset_json, outputs, data_root=None, save_stat=True, noise_files=None, snr=None, seed=0):
    """Whole-file silent-interval detection of every file of a dataset JSON (`evaluate`, M1/predict.py:38-233 with
    the prediction-phase items of M1/tools.py:297-332 and M1/dataset.py:226-252): one item per file, the whole
    recording at 14 kHz -> STFT -> net(s, v_num_frames=len(bits)) -> sigmoid -> >= 0.5.  Returns the stat dict and
    writes <outputs>/eval_results<suffix>.json.
    `noise_files` + `snr` select the clean-recordings branch (clean_audio=True): the recording is silenced on its
    labelled silent intervals, a crop of a noise file is mixed in at `snr` dB (peak 0.5) before detection, and the
    crop + its bookkeeping go to <outputs>/noise_snr<snr>/ (M1/predict.py:82-104) for create_data_from_prediction.
    The reference draws noise file and crop from Python's `random` stream; here a seeded numpy generator does
    (the draws themselves are not reproducible across the two)."""
    with open(dataset_json, 'r') as fp:
        ds = json.load(fp)
    net.eval()
    stat = []
    clean_audio = noise_files is not None
    if clean_audio and snr is None:
        raise ValueError("the clean-recordings branch needs an snr")
    suffix = convert_snr_to_suffix2(snr) if clean_audio else ''
    rng = np.random.default_rng(seed)
    noise_entries = OrderedDict()
    for data_id, f in enumerate(ds['files']):
        bits_full = f[BITSTREAM_JSON_LABEL]
        i1, i2 = _trim_unknown(bits_full)
        label = [str(int(b)) for b in bits_full[i1:i2]]
        snd, _ = audio_io.load_device(_resolve(f['audio_path'], ds.get('dataset_path'), data_root), sr=DATA_REQUIRED_SR)
        if clean_audio:
            gt = torch.tensor([int(b) for b in label], dtype=torch.uint8, device=snd.device).reshape(1, -1)
            gmask = tools.bits_to_mask_batch(gt, float(DATA_REQUIRED_SR) / f['framerate'], snd.numel())
            audio = (snd * (1 - gmask[0])).cpu().numpy()            # M1/dataset.py:247-249
            noise, _ = audio_io.load(noise_files[int(rng.integers(len(noise_files)))], sr=DATA_REQUIRED_SR)
            need = int(np.ceil(f['duration'])) * DATA_REQUIRED_SR
            start = int(rng.integers(0, max(len(noise) - need, 0) + 1))
            crop = noise[start:start + need]                        # M1/dataset.py:141-142
            mixed, _, _ = add_noise_to_audio(audio, crop, snr, start_pos=int(i1 / f['framerate'] * DATA_REQUIRED_SR))
            snd = torch.from_numpy(np.ascontiguousarray(mixed, dtype=np.float32)).to(snd.device)
            base = os.path.basename(f['path'])
            noise_name = base.split('.mp4')[0].split('.wav')[0] + '_noise.wav'
            noise_dir = os.path.join(os.path.abspath(outputs), 'noise' + suffix)
            ensure_dir(noise_dir)
            audio_io.write_wav(os.path.join(noise_dir, noise_name), crop.astype(np.float32), DATA_REQUIRED_SR)
            noise_entries[base] = OrderedDict([('audio', base.split('.mp4')[0].split('.wav')[0] + '.wav'),
                                               ('noise', noise_name), ('snr', snr)])
        S = transform.stft_batch(snd.reshape(1, -1))
        logits = net(s=S, v_num_frames=len(label))
        pred, conf = tools.threshold_bits(logits, SIGMOID_THRESHOLD)
        pred_label = [str(int(b)) for b in pred[0].cpu().numpy()]
        stat.append(OrderedDict([
            ('id', data_id), ('path', f['path']), ('full_bit_stream', bits_full), ('num_frames', f['num_frames']),
            ('framerate', f['framerate']), ('audio_sample_rate', f['audio_sample_rate']),
            ('audio_samples', f['audio_samples']), ('duration', f['duration']), ('frame_start_idx', i1),
            ('label', label), ('pred_label', pred_label), ('match', label == pred_label),
            ('confidence', [str(c) for c in conf[0].cpu().numpy()])]))
    stat_dict = OrderedDict([
        ('data_total_frames', CLIP_FRAMES), ('data_center_frames', SILENT_CONSECUTIVE_FRAMES),
        ('sigmoid_threshold', SIGMOID_THRESHOLD), ('snr', snr if clean_audio else None),
        ('prediction_statistics', OrderedDict([('all', show_metrics([b for it in stat for b in it['label']],
                                                                    [b for it in stat for b in it['pred_label']]))]))])
    stat_dict['data'] = sorted(stat, key=lambda x: np.mean([float(c) for c in x['confidence']]), reverse=True)
    if clean_audio:
        with open(os.path.join(os.path.abspath(outputs), 'noise' + suffix, suffix[1:] + '.json'), 'w') as fp:
            json.dump(OrderedDict([('snrs', [snr]), ('files', noise_entries)]), fp, **JSON_DUMP_PARAMS)
    if save_stat:
        ensure_dir(os.path.abspath(outputs))
        with open(os.path.join(os.path.abspath(outputs), 'eval_results' + suffix + '.json'), 'w') as fp:
            json.dump(stat_dict, fp, **JSON_DUMP_PARAMS)
    return stat_dict


def create_data_from_prediction(input_json, output_json=None, suffix="", noise_snr=None, save_results=True,
                                data_root=None, clean_audio=False):
    """eval_results.json -> pred_data.json (`create_data_from_prediction_newtarget_bceloss_no_voting`,
    M1/create_data_from_pred.py:38-271): per file the ground-truth, predicted and `recovered_prediction` bit
    streams; with save_results the 14 kHz signal is written to recovered<suffix>/<name>_mixed.wav next to the JSON and
    referenced as `mixed_audio`.  clean_audio=True (:158-190): the recording is mixed with the noise crop that
    detect_files stored (noise<nsuffix>/<nsuffix[1:]>.json) at its SNR, and `<name>_mixed / _clean / _full_noise.wav`
    are written and referenced (`mixed_audio`, `clean_audio`, `full_noise`, `audio_path`)."""
    suffix = suffix or ""
    if output_json is None:
        output_json = os.path.join(get_parent_dir(input_json), 'pred_data.json')
        if suffix:
            output_json = output_json.split('.json')[0] + '{}.json'.format(suffix)
    nsuffix = convert_snr_to_suffix2(noise_snr)
    output_json = output_json.split('.json')[0] + '{}.json'.format(nsuffix)
    with open(input_json, 'r') as fp:
        obj = json.load(fp)
    items = sorted(obj['data'], key=itemgetter('id'))
    groups = []
    for path, g in groupby(items, itemgetter('path')):
        g = list(g)
        groups.append(OrderedDict([
            ('path', path), ('num_frames', g[0]['num_frames']), ('framerate', g[0]['framerate']),
            ('audio_sample_rate', g[0]['audio_sample_rate']), ('audio_samples', g[0]['audio_samples']),
            ('duration', g[0]['duration']), ('bit_stream', g[0]['full_bit_stream']),
            ('ground_truth_bit_stream', ''.join(str(int(b)) for it in g for b in it['label'])),
            ('predicted_bit_stream', ''.join(str(int(b)) for it in g for b in it['pred_label'])),
            ('recovered_prediction', None), ('overlay_original', None), ('overlay_predicted', None)]))
    ds_path, labels, pred_labels = '', [], []
    # the JSON carries the authors' absolute paths: `data_root` stands in for their common directory
    src_root = os.path.commonpath([os.path.dirname(g['path']) for g in groups]) if groups else ''
    for item in groups:
        ds_path = item['path'] if ds_path == '' else find_common_path(ds_path, item['path'])
        item['num_frames'] = len(item['bit_stream'])
        item['recovered_prediction'] = item['predicted_bit_stream']
        labels += [int(s) for s in item['bit_stream']]
        pred_labels += [int(s) for s in item['recovered_prediction']]
        if save_results:
            save_dir = os.path.join(get_parent_dir(input_json), 'recovered' + suffix + nsuffix)
            ensure_dir(save_dir)
            parts = item['path'].split('.mp4')
            wav_path = parts[0] if len(parts) == 1 else parts[0] + '.wav'
            snd, _ = audio_io.load(_resolve(wav_path, src_root, data_root), sr=DATA_REQUIRED_SR)
            filename = os.path.basename(wav_path).split('.wav')[0]
            mixed_path = os.path.join(save_dir, filename + '_mixed.wav')
            rel = lambda q: os.path.join(os.path.basename(save_dir), os.path.basename(q))     # noqa: E731
            if clean_audio:
                noise_dir = os.path.join(get_parent_dir(output_json), 'noise' + nsuffix)
                with open(os.path.join(noise_dir, nsuffix[1:] + '.json'), 'r') as fpn:
                    noise_files = json.load(fpn)['files']
                entry = noise_files[os.path.basename(item['path'])]
                noise, _ = audio_io.load(os.path.join(noise_dir, entry['noise']), sr=DATA_REQUIRED_SR)
                mixed, clean, full_noise = add_noise_to_audio(snd, noise, entry['snr'], start_pos=0, norm=0.5)
                clean_path = os.path.join(save_dir, filename + '_clean.wav')
                full_noise_path = os.path.join(save_dir, filename + '_full_noise.wav')
                audio_io.write_wav(mixed_path, mixed.astype(np.float32), DATA_REQUIRED_SR)
                audio_io.write_wav(clean_path, clean.astype(np.float32), DATA_REQUIRED_SR)
                audio_io.write_wav(full_noise_path, full_noise[0].astype(np.float32), DATA_REQUIRED_SR)
                item['mixed_audio'], item['clean_audio'], item['full_noise'] = rel(mixed_path), rel(clean_path), rel(full_noise_path)
                item['audio_path'] = clean_path
            else:
                audio_io.write_wav(mixed_path, snd, DATA_REQUIRED_SR)
                item['mixed_audio'] = rel(mixed_path)
    hierarchy = OrderedDict([
        ('dataset_path', ds_path), ('num_videos', len(groups)), ('data_total_frames', obj['data_total_frames']),
        ('data_center_frames', obj['data_center_frames']), ('sigmoid_threshold', obj['sigmoid_threshold']),
        ('snr', noise_snr), ('prediction_statistics', show_metrics(labels, pred_labels)), ('files', groups)])
    with open(output_json, 'wb') as fp:
        fp.write(json.dumps(hierarchy, **JSON_DUMP_PARAMS).encode())
    return output_json


# ------------------------------------------------------------------------------------------- JSON -> model 2
def get_data_from_first_model(first_model_json_path, sr=DATA_REQUIRED_SR, snr=None, n_fft=510, hop_length=158,
                              win_length=400, unknown_clean_signal=True):
    """M2/predict.py:255-374: per file of pred_data.json load `mixed_audio`, turn `recovered_prediction` into the
    sample mask, noise_sig = mixed * mask, STFT both.  With unknown_clean_signal=False the file entries also name
    `clean_audio` and `full_noise` (written by the reference's create_data_from_pred.py with clean_audio=True): the
    clean signal is silenced on the ground-truth silent intervals (:321) and both are transformed too.
    Tensors stay in HBM: item['mixed'] / item['noise'] (/ 'clean' / 'full_noise') are (1, 2, 256, T) GPU tensors,
    item['mask'] a (n_samples,) GPU tensor."""
    with open(first_model_json_path, 'r') as fp:
        obj = json.load(fp)
    snr = obj['snr']
    data_list = []
    for data in obj['files']:
        mixed_audio_path = os.path.join(get_parent_dir(first_model_json_path), data['mixed_audio'])
        mixed_sig, _ = audio_io.load_device(mixed_audio_path, sr=sr)
        bitstream = data[BIT_STREAM_LABEL]
        vals = []
        for bit in bitstream:                      # M2/predict.py:232-252: '0' / '1' (a '2' is tolerated as non-silent)
            if bit not in '012':
                print('Invalid bit?')
                raise RuntimeError
            vals.append(0 if bit == '0' else 1)
        bits = torch.tensor(vals, dtype=torch.uint8, device=mixed_sig.device).reshape(1, -1)
        mask, noise_sig = tools.bits_to_mask_batch(bits, float(sr) / data['framerate'], mixed_sig.numel(),
                                                   mixed_sig.reshape(1, -1))
        item = OrderedDict([('id', os.path.splitext(os.path.basename(data['path']))[0]), ('path', data['path'])])
        known = OrderedDict()
        if not unknown_clean_signal:
            clean_audio_path = os.path.join(get_parent_dir(first_model_json_path), data['clean_audio'])
            full_noise_path = os.path.join(get_parent_dir(first_model_json_path), data['full_noise'])
            clean_sig, _ = audio_io.load_device(clean_audio_path, sr=sr)
            full_noise, _ = audio_io.load_device(full_noise_path, sr=sr)
            gt = torch.tensor([0 if b == '0' else 1 for b in data[GT_BIT_STREAM_LABEL]], dtype=torch.uint8,
                              device=clean_sig.device).reshape(1, -1)
            gt_mask = tools.bits_to_mask_batch(gt, float(sr) / data['framerate'], clean_sig.numel())
            clean_sig = clean_sig * (1 - gt_mask[0])             # silent intervals truly silent (M2/predict.py:321)
            item['clean_audio_path'] = clean_audio_path
            known['clean'] = transform.stft_batch(clean_sig.reshape(1, -1), n_fft, hop_length, win_length)
            known['full_noise'] = transform.stft_batch(full_noise.reshape(1, -1), n_fft, hop_length, win_length)
        item['mixed_audio_path'] = mixed_audio_path
        if not unknown_clean_signal:
            item['full_noise_path'] = full_noise_path
        item['bitstream'] = bitstream
        item['mixed'] = transform.stft_batch(mixed_sig.reshape(1, -1), n_fft, hop_length, win_length)
        if 'clean' in known:
            item['clean'] = known['clean']
        item['noise'] = transform.stft_batch(noise_sig, n_fft, hop_length, win_length)
        if 'full_noise' in known:
            item['full_noise'] = known['full_noise']
        item.update([('mask', mask[0]), ('snr', snr), ('sr', sr)])
        data_list.append(item)
    info = OrderedDict([(k, obj[k]) for k in ('dataset_path', 'num_videos', 'data_total_frames', 'data_center_frames',
                                              'sigmoid_threshold')])
    return data_list, info


@torch.no_grad()
def denoise_files(net, data_list_info, outputs, snr=None, threshold="", save_individual_results=True, save_stat=True,
                  pesq_fn=None, stoi_fn=None):
    """M2/predict.py:377-576 (`evaluate`): net(mixed, noise) -> (pred_noise, mask); mask applied to the mixed
    spectrogram; ISTFT of mixed / noise intervals / predicted noise / output written as
    <outputs>/<snr suffix>/<id>/*.wav + stat.json, and <outputs>/eval_results<suffixes>.json.  Items that carry a
    `clean` spectrogram (unknown_clean_signal=False) also get the objective measures of the output resampled to
    16 kHz against the clean signal (:455-460) and the ground-truth WAVE files; `pesq_fn(clean, output, sr)` /
    `stoi_fn(clean, output, sr)` supply the two third-party scores (None entries otherwise) and the averages of the
    available measures go to `denoise_statistics`."""
    data_list, data_info = data_list_info
    data_info = OrderedDict(data_info)
    data_info['snr'] = snr
    net.eval()
    stat = []
    for data in data_list:
        pred_noise_stft, crm = net(data['mixed'], data['noise'])
        out_stft = transform.batch_fast_icRM_sigmoid(data['mixed'], crm)
        sigs = transform.istft_batch(torch.cat([data['mixed'], data['noise'], pred_noise_stft, out_stft], dim=0))
        known = 'clean' in data
        info = OrderedDict([('id', str(data['id'])), ('path', str(data['path']))])
        if known:
            info['clean_audio_path'] = data['clean_audio_path']
        info['mixed_audio_path'] = data['mixed_audio_path']
        if known:
            info['full_noise_path'] = data['full_noise_path']
        info.update([('bitstream', data['bitstream']), ('sr', data['sr']), ('snr', data['snr'])])
        gt_sigs = None
        if known:
            gt_sigs = transform.istft_batch(torch.cat([data['full_noise'], data['clean']], dim=0))
            out16 = audio_io.resample_device(sigs[3].contiguous(), data['sr'], 16000)
            clean16 = audio_io.resample_device(gt_sigs[1].contiguous(), data['sr'], 16000)
            pesq = pesq_fn(clean16.cpu().numpy(), out16.cpu().numpy(), 16000) if pesq_fn is not None else None
            stoi = stoi_fn(clean16.cpu().numpy(), out16.cpu().numpy(), 16000) if stoi_fn is not None else None
            info.update(metrics.evaluate_metrics(out16, clean16, sr=16000, pesq=pesq, stoi=stoi))
        if save_individual_results:
            save_dir = os.path.join(os.path.abspath(outputs), convert_snr_to_suffix2(snr)[1:], str(data['id']))
            ensure_dir(save_dir)
            host = sigs.cpu().numpy()
            for k, name in enumerate(('noisy_input', 'noise_intervals', 'predicted_full_noise', 'denoised_output')):
                p = os.path.join(save_dir, name + '.wav')
                audio_io.write_wav(p, host[k], data['sr'])
                info[name] = p
            if known:
                gh = gt_sigs.cpu().numpy()
                for k, name in enumerate(('ground_truth_full_noise', 'ground_truth_clean_input')):
                    p = os.path.join(save_dir, name + '.wav')
                    audio_io.write_wav(p, gh[k][:host.shape[1]], data['sr'])
                    info[name] = p
            with open(os.path.join(save_dir, 'stat.json'), 'w') as fp:
                json.dump(info, fp, **JSON_DUMP_PARAMS)
        stat.append(info)
    if save_stat:
        if stat and 'l1' in stat[0]:
            keys = ('l1', 'stoi', 'csig', 'cbak', 'covl', 'pesq', 'ssnr_regular', 'ssnr_shift', 'ssnr_clip', 'ssnr_exsi', 'overall_snr')
            data_info['denoise_statistics'] = OrderedDict(
                ('avg_' + k, (sum(it[k] for it in stat) / len(stat)) if all(it[k] is not None for it in stat) else None)
                for k in keys)
        data_info['files'] = stat
        ensure_dir(os.path.abspath(outputs))
        path = os.path.join(os.path.abspath(outputs), 'eval_results' + convert_threshold_to_suffix(threshold) +
                            convert_snr_to_suffix2(snr) + '.json')
        with open(path, 'w') as fp:
            json.dump(data_info, fp, **JSON_DUMP_PARAMS)
    return stat
