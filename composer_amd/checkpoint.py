"""Checkpoint directory contract of the reference train loop (transformer.py:890-891,941-943,953-955;
models/__init__.py:66-90): `<logdir>/{checkpoint, ckpt-N.*}` with a `checkpoint` text pointer, at most
`max_to_keep` checkpoints kept, `latest_checkpoint` resolution.

Payload, by `format`: 'npz' (default) = one `ckpt-N.npz` holding the tensors of the object graph (model/<param>,
optimizer/{m,v}/<param>, optimizer/iter) and step / epoch / save_counter; 'tensorbundle' = the reference's own
`ckpt-N.index` + `ckpt-N.data-00000-of-00001` pair (composer_amd/tensorbundle.py; not checked against TensorFlow here).
`load` reads either, whichever exists.  Plain host code: no GPU involved.
"""
import json
import os
import re
from pathlib import Path

import numpy as np


class CheckpointManager:
    """tf.train.CheckpointManager(checkpoint, directory, max_to_keep) semantics."""

    def __init__(self, directory, max_to_keep=1, format='npz'):
        if format not in ('npz', 'tensorbundle'):
            raise ValueError("checkpoint format must be 'npz' or 'tensorbundle'")
        self.directory = Path(directory) if directory is not None else None
        self.max_to_keep = max_to_keep
        self.format = format
        self._paths = []
        self._counter = 0
        if self.directory is not None and (self.directory / 'checkpoint').exists():
            txt = (self.directory / 'checkpoint').read_text()
            self._paths = [self.directory / p for p in re.findall(r'all_model_checkpoint_paths: "([^"]+)"', txt)]
            self._paths = [p for p in self._paths if _exists(p)]
            for p in self._paths:
                m = re.search(r'ckpt-(\d+)$', str(p))
                if m:
                    self._counter = max(self._counter, int(m.group(1)))

    @property
    def latest_checkpoint(self):
        return str(self._paths[-1]) if self._paths else None

    @property
    def checkpoints(self):
        return [str(p) for p in self._paths]

    def save(self, tensors, meta):
        self.directory.mkdir(parents=True, exist_ok=True)
        self._counter += 1                                   # save_counter
        prefix = self.directory / ('ckpt-%d' % self._counter)
        meta = dict(meta, save_counter=self._counter)
        if self.format == 'tensorbundle':
            from . import tensorbundle
            tensorbundle.write_bundle(prefix, tensorbundle.bundle_from_state(tensors, meta))
        else:
            payload = dict(tensors)
            payload['__meta__'] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
            tmp = str(prefix) + '.tmp.npz'
            np.savez(tmp, **payload)
            os.replace(tmp, str(prefix) + '.npz')
        self._paths.append(prefix)
        if self.max_to_keep:
            while len(self._paths) > self.max_to_keep:
                old = self._paths.pop(0)
                for ext in _EXTS:
                    try:
                        os.remove(str(old) + ext)
                    except OSError:
                        pass
        lines = ['model_checkpoint_path: "%s"' % self._paths[-1].name]
        lines += ['all_model_checkpoint_paths: "%s"' % p.name for p in self._paths]
        (self.directory / 'checkpoint').write_text('\n'.join(lines) + '\n')
        return str(prefix)


_EXTS = ('.npz', '.index', '.data-00000-of-00001')


def _exists(prefix):
    return Path(str(prefix) + '.npz').exists() or Path(str(prefix) + '.index').exists()


def load(prefix):
    if prefix is None:
        raise FileNotFoundError('no checkpoint to restore')
    if not Path(str(prefix) + '.npz').exists() and Path(str(prefix) + '.index').exists():
        from . import tensorbundle
        return tensorbundle.state_from_bundle(tensorbundle.read_bundle(prefix))
    with np.load(str(prefix) + '.npz') as z:
        meta = json.loads(bytes(z['__meta__']).decode()) if '__meta__' in z.files else {}
        tensors = {k: z[k] for k in z.files if k != '__meta__'}
    return tensors, meta


class ScalarLog:
    """tf.summary.create_file_writer + tf.summary.scalar (transformer.py:903,933-951): same scalar names and steps, written
    twice into `<logdir>/train/`: a TensorBoard event file (composer_amd/tbevents.py) and `scalars.jsonl` (JSON lines)."""

    def __init__(self, directory):
        from composer_amd import tbevents
        self.directory = Path(directory)
        self.directory.mkdir(parents=True, exist_ok=True)
        self._f = open(self.directory / 'scalars.jsonl', 'a')
        self._events = tbevents.EventFileWriter(self.directory)

    def scalar(self, name, value, step):
        self._f.write(json.dumps({'tag': name, 'value': float(value), 'step': int(step)}) + '\n')
        self._events.scalar(name, value, step)

    def flush(self):
        self._f.flush()
        self._events.flush()

    def close(self):
        self._f.close()
        self._events.close()
