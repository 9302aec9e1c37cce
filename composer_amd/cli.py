"""`composer train | evaluate | generate | make-config | summary` for the Transformer hot path.

Mirrors the commands, arguments and defaults of the reference's composer/cli.py (train :516-589, evaluate :591-615,
generate :617-680, make-config :69-78, summary :424-440) with the TensorFlow model replaced by the HIP one.  Out of
scope: the MusicRNN model type, MIDI preprocessing/synthesis.  `export-dataset` (:346-380) and
`.tfrecord` dataset paths (:232-268) go through composer_amd.tfrecord.
Documented divergences: `--temperature 0` means greedy argmax (the reference divides by the temperature, cli.py:671);
besides the reference's MIDI prompt (`--prompt`, read by composer_amd.midi instead of pretty_midi) a prompt can be given
as event ids (`--prompt-ids` / `--prompt-data`), and an output path ending in `.data` gets event ids instead of a MIDI
file; `--decode-mode` selects the literal loop of cli.py:663-676 or a real KV cache."""
import datetime
import logging
import os
import shutil
import sys
from enum import Enum, unique
from pathlib import Path

import click
import numpy as np

from . import config as cfgmod
from . import dataset as ds
from . import tfrecord
from .transformer import ModelSaveFrequencyMode, Transformer


@unique
class ModelType(Enum):
    MUSIC_RNN = 'music_rnn'
    TRANSFORMER = 'transformer'


class EnumType(click.Choice):
    """Case-insensitive enum names, reference composer/click_utils.py:10-82."""

    def __init__(self, enum, case_sensitive=False):
        self._enum = enum
        super().__init__([e.name.lower() for e in enum] + [e.value for e in enum], case_sensitive=case_sensitive)

    def convert(self, value, param, ctx):
        if isinstance(value, self._enum):
            return value
        v = super().convert(value, param, ctx).lower()
        for e in self._enum:
            if v in (e.name.lower(), str(e.value).lower()):
                return e
        self.fail('invalid choice: %s' % value, param, ctx)


def get_default_config():
    return Path(__file__).parent / 'default_config.yml'


def _require_transformer(model_type):
    if model_type != ModelType.TRANSFORMER:
        logging.error('Only the transformer model type is implemented on MI355X (the MusicRNN is outside the hot path).')
        sys.exit(1)


def _runtime(config):
    rt = config.transformer.get('runtime', None) or {}
    return rt.get('dtype', 'bf16'), int(rt.get('seed', 0))


def _vocab(config):
    d = config.dataset
    return ds.vocab_size(d.time_step_increment, d.max_time_steps, d.velocity_bins)


def create_model(model_type, config, **kwargs):
    """cli.py:95-141: positional constructor arguments in the reference's order."""
    _require_transformer(model_type)
    m = config.transformer.model
    dtype, seed = _runtime(config)
    vocab = _vocab(config)
    model = Transformer(
        vocab, m.embedding_size, m.window_size, m.decoder_layers_count, m.attention_head_count,
        m.use_relative_attention, m.initializer_mean, m.initializer_stddev, m.attention_dropout_rate,
        m.residual_dropout_rate, m.layer_normalization_epsilon, m.scale_attention, m.use_layer_normalization,
        dtype=kwargs.get('dtype', dtype), seed=seed, max_batch=kwargs.get('max_batch', config.transformer.train.batch_size),
        max_seq=m.window_size)
    return model, vocab


def get_dataset(model_type, dataset_path, config, mode='', max_files=None, shuffle_files=True, shuffle_dataset=True,
                rank=0, world_size=1, seed=0):
    """cli.py:185-276: a directory dataset (`<root>/{train,test}/**/*.data`) or an exported `.tfrecord` file."""
    if mode not in ('train', 'test', ''):
        raise ValueError('\'{}\' is an invalid dataset mode! Must be one of: \'train\', \'test\', or none.'.format(mode))
    p = Path(dataset_path)
    if not p.is_dir():
        if not p.is_file() or p.suffix != '.tfrecord':                                   # cli.py:232-241
            logging.error('\'{}\' is an invalid dataset path! The dataset can either be a directory of processed MIDI '
                          'files or a TFRecord file.'.format(p))
            sys.exit(1)
        dataset, header = tfrecord.load_tfrecord_dataset(p, shuffle=shuffle_dataset, seed=seed, rank=rank, world_size=world_size)
        warning = 'The TFRecord file was probably exported using a different config.'    # cli.py:246-268
        if header['model_type'] != model_type.value:
            logging.warning('Model type mismatch when loading \'{}\'. Expected {} but found {}. {}'.format(
                p, model_type.value, header['model_type'], warning))
            click.confirm('Do you want to continue? This may cause errors or corrupt the training session.', abort=True)
        for name, want in (('batch', config.transformer.train.batch_size), ('window', config.transformer.model.window_size)):
            if header[name + '_size'] != want:
                logging.error('Expected a {} size of {} but found {}. {}'.format(name, want, header[name + '_size'], warning))
                sys.exit(1)
        return dataset
    p = p / mode
    if not p.exists():
        logging.error('Could not get {} dataset since \'{}\' has no {} folder.'.format(mode, dataset_path, mode))
        sys.exit(1)
    files = ds.get_processed_files(p)
    if shuffle_files:
        np.random.default_rng(seed).shuffle(files)          # cli.py:229-230 (np.random.shuffle, unseeded there)
    if max_files is not None:
        files = files[:max_files]
    d = config.dataset
    return ds.load_dataset(files, config.transformer.train.batch_size, config.transformer.model.window_size,
                           shuffle=shuffle_dataset, seed=seed, rank=rank, world_size=world_size,
                           expect_settings=(d.time_step_increment, d.max_time_steps, d.velocity_bins))


def get_config_from_restoredir(restoredir):
    """cli.py:500-514"""
    path = Path(restoredir) / 'config.yml'
    if not path.exists():
        logging.error('Failed to restore model from \'{}\'! Could not find \'config.yml\' file!'.format(restoredir))
        sys.exit(1)
    return cfgmod.get(path)


def _init_distributed(model):
    """One process per GPU under torch.distributed.run: gloo for the bootstrap, RCCL (inside the library) for gradients."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if not dist.is_initialized():
            dist.init_process_group('gloo', rank=rank, world_size=world)
        uid = [Transformer.new_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        model.init_data_parallel(rank, world, uid[0])
    return rank, world


@click.group()
@click.option('--verbosity', '-v', default='info', help='Logging level name.')
def cli(verbosity):
    if int(os.environ.get('WORLD_SIZE', '1')) > 1:
        # Library load order: a data-parallel job uses torch.distributed (gloo) to hand the RCCL id to the ranks, so
        # torch comes into this process anyway -- import it BEFORE libcomposer_hip.so is loaded, so that the dynamic linker binds the
        # library to the HIP runtime and the RCCL torch has already mapped: one runtime, the RCCL bench.py and the tests run on.
        import torch.distributed  # noqa: F401
    logging.basicConfig(level=getattr(logging, str(verbosity).upper(), logging.INFO), format='%(levelname)s: %(message)s')


@cli.command('make-config')
@click.argument('filepath', default='config.yml')
def make_config(filepath):
    """Creates a configuration file from the default configuration (cli.py:69-78)."""
    shutil.copy(get_default_config(), filepath)
    logging.info('Created configuration file \'{}\'.'.format(filepath))


@cli.command()
@click.argument('model-type', type=EnumType(ModelType, False))
@click.option('-c', '--config', 'config_filepath', default=None)
def summary(model_type, config_filepath):
    """Prints the parameter table of the model (cli.py:424-440)."""
    config = cfgmod.get(config_filepath or get_default_config())
    model, _ = create_model(model_type, config)
    model.summary()


@cli.command()
@click.argument('model-type', type=EnumType(ModelType, False))
@click.argument('dataset-path')
@click.option('--logdir', default='./output/logdir/', help='The root log directory. Defaults to \'./output/logdir\'.')
@click.option('--restoredir', default=None, type=str, help='The directory of the model to continue training.')
@click.option('-c', '--config', 'config_filepath', default=None)
@click.option('-e', '--epochs', 'epochs', default=10, help='The number of epochs to train for. Defaults to 10.')
@click.option('--use-generator/--no-use-generator', default=False, help='Accepted for compatibility; datasets are memory-mapped ids either way.')
@click.option('--max-files', default=None, type=int)
@click.option('--save-freq-mode', 'save_frequency_mode', type=EnumType(ModelSaveFrequencyMode, False), default='global_step')
@click.option('--save-freq', 'save_frequency', type=int, default=500)
@click.option('--max-checkpoints', 'max_checkpoints', type=int, default=3)
@click.option('--show-progress-bar/--no-show-progress-bar', 'show_progress_bar', default=True)
@click.option('--max-steps', default=None, type=int, help='Stop after this many steps (not in the reference; for smoke runs).')
@click.option('--checkpoint-format', type=click.Choice(['npz', 'tensorbundle']), default='npz',
              help='npz (default) or the reference\'s TensorBundle files (ckpt-N.index / .data-00000-of-00001). Either is restored.')
def train(model_type, dataset_path, logdir, restoredir, config_filepath, epochs, use_generator, max_files,
          save_frequency_mode, save_frequency, max_checkpoints, show_progress_bar, max_steps, checkpoint_format):
    """Trains the specified model (cli.py:516-589)."""
    _require_transformer(model_type)
    rank = int(os.environ.get('RANK', '0'))
    if restoredir is not None:
        config = get_config_from_restoredir(restoredir)
        model_logdir = None
    else:
        stamp = datetime.datetime.now().strftime('%Y-%m-%d_%H-%M-%S')
        model_logdir = Path(logdir) / '{}-{}'.format(model_type.name.lower(), stamp)
        config = cfgmod.get(config_filepath or get_default_config())
        if rank == 0:
            model_logdir.mkdir(parents=True, exist_ok=True)
            banner = '\n'.join([
                '#########################################################',
                '# Datetime: {}.'.format(datetime.datetime.now()),
                '#########################################################',
                '# This is an autogenerated backup of the configuration file',
                '# used when invoking the train command.',
                '# ',
                '# DO NOT MODIFY THIS FILE!',
                '# Doing so may cause errors upon resuming training.',
                '#########################################################'])
            with open(config.filepath) as src, open(model_logdir / 'config.yml', 'w+') as dst:
                dst.write(banner + '\n' + src.read())
    model, _ = create_model(model_type, config)
    rank, world = _init_distributed(model)
    _, seed = _runtime(config)
    dataset = get_dataset(model_type, dataset_path, config, 'train', max_files=max_files, rank=rank, world_size=world, seed=seed)
    input_shape = (config.transformer.train.batch_size, config.transformer.model.window_size)
    model.train(dataset, input_shape, model_logdir, restoredir=restoredir, epochs=epochs,
                learning_rate=config.transformer.train.learning_rate, save_frequency_mode=save_frequency_mode,
                save_frequency=save_frequency, max_checkpoints=max_checkpoints,
                show_progress_bar=show_progress_bar and rank == 0, max_steps=max_steps, checkpoint_format=checkpoint_format)
    if rank == 0 and model_logdir is not None:
        click.echo(str(model_logdir))


@cli.command('export-dataset')
@click.argument('model-type', type=EnumType(ModelType, False))
@click.argument('preprocessed-path')
@click.argument('output-path')
@click.option('-c', '--config', 'config_filepath', default=None)
@click.option('--use-generator/--no-use-generator', default=False, help='Accepted for compatibility.')
@click.option('--max-files', default=None, type=int)
def export_dataset(model_type, preprocessed_path, output_path, config_filepath, use_generator, max_files):
    """Exports a processed dataset input pipeline as a TFRecord file (cli.py:346-380): the unshuffled batches of the
    `.data` files under PREPROCESSED-PATH, windowed and batched by the config."""
    _require_transformer(model_type)
    config = cfgmod.get(config_filepath or get_default_config())
    dataset = get_dataset(model_type, preprocessed_path, config, shuffle_dataset=False, max_files=max_files)
    logging.info('Loading dataset and writing to TFRecord. This make take a while...')
    n = tfrecord.export_dataset(dataset, output_path, model_type.value)
    logging.info('Finished exporting \'{}\' as a TFRecord: \'{}\' ({} batches)'.format(preprocessed_path, output_path, n))


@cli.command()
@click.argument('model-type', type=EnumType(ModelType, False))
@click.argument('dataset-path')
@click.argument('restoredir')
@click.option('--use-generator/--no-use-generator', default=False)
@click.option('--max-files', default=None, type=int)
def evaluate(model_type, dataset_path, restoredir, use_generator, max_files):
    """Evaluate the specified model (cli.py:591-615)."""
    config = get_config_from_restoredir(restoredir)
    model, _ = create_model(model_type, config)
    model.load_from_checkpoint(restoredir)
    model.compile(config.transformer.train.learning_rate)
    model.build(input_shape=(config.transformer.train.batch_size, None))
    test = get_dataset(model_type, dataset_path, config, 'test', max_files=max_files, shuffle_dataset=False)
    loss, accuracy = model.evaluate(test, verbose=0)
    logging.info('- Finished evaluating model. Loss: {:.4f}, Accuracy: {:.4f}'.format(loss, accuracy))
    click.echo('loss {:.6f} accuracy {:.6f}'.format(loss, accuracy))


@cli.command()
@click.argument('model-type', type=EnumType(ModelType, False))
@click.argument('restoredir')
@click.argument('output-filepath')
@click.option('--prompt', '-p', 'prompt', default=None, help='The path of the MIDI file to prompt the network with.')
@click.option('--prompt-ids', default=None, help='Comma-separated event ids to prompt the network with (instead of --prompt).')
@click.option('--prompt-data', default=None, help='A .data file whose first events prompt the network (instead of --prompt).')
@click.option('--prompt-length', default=10, help='Number of events to take from the start of the prompt. Defaults to 10.')
@click.option('--length', '-l', 'generate_length', default=1024, help='The length of the generated event sequence. Defaults to 1024')
@click.option('--temperature', default=1.0, help='Sampling temperature; 0 = greedy argmax. Defaults to 1.0.')
@click.option('--decode-mode', type=click.Choice(['reference-literal', 'kv-cache']), default=None,
              help='kv-cache: model(x, past=presents); reference-literal: the reference\'s loop as written (no past). '
                   'Default: kv-cache when prompt + length fits window_size, else reference-literal.')
def generate(model_type, restoredir, output_filepath, prompt, prompt_ids, prompt_data, prompt_length, generate_length,
             temperature, decode_mode):
    """Generate a MIDI file (cli.py:617-680): MIDI prompt -> event ids -> model -> event ids -> MIDI.  An output path
    ending in `.data` gets the event ids in the dataset's binary format instead of a MIDI file."""
    from composer_amd import notes as nt
    config = get_config_from_restoredir(restoredir)
    model, _ = create_model(model_type, config, dtype='fp32')
    model.load_from_checkpoint(restoredir)
    model.compile(config.transformer.train.learning_rate)
    model.build(input_shape=(1, None))
    d = config.dataset
    if prompt is not None:                                       # cli.py:645-660
        x = nt.prompt_ids_from_midi(prompt, prompt_length, d.time_step_increment, d.max_time_steps, d.velocity_bins)
    elif prompt_ids is not None:
        x = [int(t) for t in prompt_ids.split(',') if t.strip() != '']
    elif prompt_data is not None:
        x = ds.read_data_file(prompt_data)[0].astype(np.int32).tolist()
    else:
        raise NotImplementedError()                              # cli.py:642-643
    x = x[:prompt_length]                                        # cli.py:649
    model.reset_states()
    window = config.transformer.model.window_size
    fits = len(x) + generate_length - 1 <= window
    if decode_mode == 'kv-cache' and not fits:
        # position ids would run past the wpe table (transformer.py:675-679,786): an explicit request cannot be honoured, and
        # silently switching modes would change what is sampled (context-conditioned vs the reference's context-free loop)
        raise click.UsageError('--decode-mode kv-cache: prompt ({}) + length ({}) - 1 exceeds window_size ({}); at most '
                               '--length {} fits, or use --decode-mode reference-literal.'.format(
                                   len(x), generate_length, window, window - len(x) + 1))
    if decode_mode is None:
        # no mode asked for: the KV cache when it fits; otherwise the reference's own loop, which never feeds `past` back
        # (cli.py:663-676) and can therefore emit any length -- said on stderr whatever the log level
        decode_mode = 'kv-cache' if fits else 'reference-literal'
        if not fits:
            click.echo('composer generate: prompt ({}) + length ({}) - 1 exceeds window_size ({}); using the reference\'s '
                       'decode loop (--decode-mode reference-literal). With the KV cache at most --length {} fits.'.format(
                           len(x), generate_length, window, window - len(x) + 1), err=True)
    click.echo('decode-mode: {}'.format(decode_mode), err=True)
    ids = model.generate(x, generate_length, temperature=temperature, mode=decode_mode)
    all_ids = list(x) + ids.tolist()                             # prompt + generated (cli.py:676)
    out = Path(output_filepath)
    out.parent.mkdir(parents=True, exist_ok=True)
    if out.suffix == '.data':
        vr = ds.event_value_ranges(d.time_step_increment, d.max_time_steps, d.velocity_bins)
        rg = ds.event_ranges(vr)
        events = [ds.id_to_event(int(i), rg, vr) for i in all_ids]
        ds.write_data_file(out, events, d.time_step_increment, d.max_time_steps, d.velocity_bins)
    else:                                                        # cli.py:678-680
        nt.ids_to_midi(all_ids, out, d.time_step_increment, d.max_time_steps, d.velocity_bins)
    click.echo(','.join(str(int(i)) for i in ids))


if __name__ == '__main__':
    cli()
