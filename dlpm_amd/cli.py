"""`eval.py --generate`-compatible command line for the sampling path (no dataset / neptune needed).

Covers the reference flags that reach the sampler (script_utils.py:11-12,35-39,56-67,82-83,155-221):
  --config --method --generate --reverse_steps --deterministic --clip --alpha --non_iso --scale --input_scaling --set_seed/--random_seed
plus the checkpoint to evaluate, resolved like eval.py does (eval.py:21, bem/utils_exp.py:96-139):
  --name N [--models_dir models] [--epoch E]  ->  models/N/<dataset>/model_<exphash>[_<E>].pt
or given directly with --checkpoint FILE; --ema_eval [--ema_index I] evaluates an EMA shadow.
Samples are produced in chunks of eval.batch_size like EvaluationManager (:181-193).
"""
import argparse
import os
import sys

import numpy as np
import torch

import dlpm_amd
from dlpm_amd.config import sample_shape


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument('--config', required=True, help='config name (dlpm_amd/configs) or path to a reference-schema YAML')
    ap.add_argument('--method', default=None, choices=['dlpm', 'lim'], help='generative method (script_utils.py:6-7)')
    ap.add_argument('--generate', type=int, default=None, help='number of samples (eval.data_to_generate)')
    ap.add_argument('--reverse_steps', type=int, default=None)
    ap.add_argument('--alpha', type=float, default=None)
    ap.add_argument('--non_iso', action='store_true', help='non-isotropic noise (script_utils.py:26-27)')
    ap.add_argument('--scale', default=None, choices=['scale_preserving', 'scale_exploding'], help='script_utils.py:104-105')
    ap.add_argument('--input_scaling', action='store_true', help='script_utils.py:107-108')
    ap.add_argument('--deterministic', action='store_true', help='DLIM sampling')
    ap.add_argument('--clip', action='store_true', help='clip_denoised')
    ap.add_argument('--set_seed', type=int, default=None)
    ap.add_argument('--random_seed', action='store_true')
    ap.add_argument('--batch_size', type=int, default=None, help='eval.batch_size override')
    ap.add_argument('--checkpoint', default=None, help='reference checkpoint (.pt) with model_parameters / ema_models')
    ap.add_argument('--name', default=None, help='experiment name: checkpoints are looked up under <models_dir>/<name>/')
    ap.add_argument('--models_dir', default='models')
    ap.add_argument('--epoch', type=int, default=None, help='checkpointed epoch to load (default: the latest)')
    ap.add_argument('--ema_eval', action='store_true')
    ap.add_argument('--ema_index', type=int, default=0, help='which EMA shadow (order of training.<method>.ema_rates)')
    ap.add_argument('--synthetic_weights', type=int, default=None, metavar='SEED',
                    help='random init with the zero-initialised tensors re-drawn (benchmarks; NOT the reference init)')
    ap.add_argument('--rng', default='philox', choices=['philox', 'reference'])
    ap.add_argument('--conv', default='auto', choices=['auto', 'f4', 'f2', 'igemm'],
                    help='3x3 convolution generation of the UNet (UNetModel.set_conv_policy): auto = fastest per layer '
                         '(Winograd F(4x4,3x3) where it applies, 1.6e-5 of the reference per forward), f2 / igemm = 3e-6 / '
                         '4e-6 at 1.33x / 2.2x the time; never a function of the batch, so chunking does not change a pixel')
    ap.add_argument('--gemm', default='auto', choices=['auto', 'f32', 'bf16x3'],
                    help='matrix pipe of the 1x1 and stride-2 convolutions (UNetModel.set_gemm_policy): bf16x3 = fp32 operands split '
                         'exactly into three bf16 planes, fp32 accumulate (fp32-grade results, the default where the shape admits it); '
                         'f32 = the fp32 MFMA everywhere')
    ap.add_argument('--out', default=None, help='.npy file for the generated samples')
    ap.add_argument('--gen_data_path', default=None,
                    help='directory for <i>.png files (EvaluationManager image dump); images only')
    ap.add_argument('--device_batch', default='auto',
                    help='with --gen_data_path and --rng philox: sample in chunks of at least this many images (pixels do not '
                         'depend on the chunking).  auto (default) = up to 1024, as many as free HBM holds; 0 = eval.batch_size '
                         'chunks exactly as the reference')
    a = ap.parse_args(argv)

    p = dlpm_amd.load_config(a.config)
    p['device'] = 'cuda'
    if a.method is not None:
        p['method'] = a.method
    m = p['method']
    if a.alpha is not None:
        p[m]['alpha'] = a.alpha
    if a.non_iso:
        p[m]['isotropic'] = False
    if a.scale is not None:
        p[m]['scale'] = a.scale
    if a.input_scaling:
        p[m]['input_scaling'] = True
    if a.generate is not None:
        assert a.generate <= p['eval']['real_data'], 'cannot generate more data than the number of real data'
        p['eval']['data_to_generate'] = a.generate
    if a.reverse_steps is not None:
        p['eval'][m]['reverse_steps'] = a.reverse_steps
    if a.deterministic:
        p['eval'][m]['deterministic'] = True
    if a.clip:
        p['eval'][m]['clip_denoised'] = True
    if a.batch_size is not None:
        p['eval']['batch_size'] = a.batch_size
    seed = None if a.random_seed else a.set_seed
    if seed is not None:
        torch.manual_seed(seed)
        np.random.seed(seed)

    model = dlpm_amd.init_model_by_parameter(p)
    path = a.checkpoint
    if path is None and a.name is not None:
        path = dlpm_amd.checkpoint.find_checkpoint(p, os.path.join(a.models_dir, a.name), epoch=a.epoch)
    if path:
        epoch, steps = dlpm_amd.checkpoint.load_into(model, path, ema=a.ema_index if a.ema_eval else None)
        print('loaded %s (epoch %s, %s steps%s)' % (path, epoch, steps, ', ema #%d' % a.ema_index if a.ema_eval else ''),
              file=sys.stderr)
    elif a.synthetic_weights is not None:
        dlpm_amd.rerandomize_(model, a.synthetic_weights)
    if a.conv != 'auto':
        if not hasattr(model, 'set_conv_policy'):
            raise SystemExit('--conv applies to the UNet score networks')
        model.set_conv_policy(a.conv)
    if a.gemm != 'auto':
        if not hasattr(model, 'set_gemm_policy'):
            raise SystemExit('--gemm applies to the UNet score networks')
        model.set_gemm_policy(a.gemm)
    method = dlpm_amd.init_method_by_parameter(p, rng=a.rng, seed=seed or 0)
    is_image = dlpm_amd.is_image_dataset(p['data']['dataset'])
    gm = dlpm_amd.GenerationManager(method, dlpm_amd.ShapeProbe(sample_shape(p)), is_image, **p['eval'][m])
    if a.gen_data_path:
        assert is_image, '--gen_data_path dumps images; 2-D data has no image form'
        ev = dlpm_amd.EvaluationManager(method, gm, None, is_image=True, gen_data_path=a.gen_data_path,
                                        device_batch=a.device_batch if a.device_batch == 'auto' else int(a.device_batch))
        r = ev.evaluate_model({'default': model}, data_to_generate=p['eval']['data_to_generate'],
                              batch_size=p['eval']['batch_size'])
        print('wrote %d png files to %s' % (r['generated'], r['gen_data_path']))
        return r
    remaining, chunks = p['eval']['data_to_generate'], []
    while remaining > 0:                                    # EvaluationManager.py:181-193
        n = min(p['eval']['batch_size'], remaining)
        chunks.append(gm.generate({'default': model}, n).clone())
        remaining -= n
        print('generated %d, %d to go' % (n, remaining), file=sys.stderr)
    samples = torch.cat(chunks)
    if a.out:
        np.save(a.out, samples.numpy())
    print('samples %s  mean %.4f  min %.4f  max %.4f' % (tuple(samples.shape), samples.mean(), samples.min(), samples.max()))
    return samples


if __name__ == '__main__':
    main()
