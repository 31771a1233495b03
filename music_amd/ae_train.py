"""Counterpart of the reference's ``wavenet_autoencoder/train.py`` (training harness of the
autoencoder) for the MI355X path.

The reference file is a tab-indented copy of ``wavenet/train.py`` that cannot run as shipped: it
imports ``faster_audio_data`` and reads ``./params/train_params.json`` / ``dataset_params.json`` that
only exist in ``wavenet/``, its ``model_params.json`` is invalid JSON, ``optim.sgd`` is a
typo (:28), ``sorted(keys=...)`` raises (:159) and ``int(name[7:])`` raises for its own checkpoint
names (:77-78).  This module keeps its surface and file formats - the reference's own line formats
``"Average loss is X\\n"`` (:144-146) and ``"Epoch{N}model saved!"`` (:164-166, no separators), checkpoints
``wavenet_autoencoder{N}.model`` - and fixes only what cannot work (the resume counter, which the reference parses out
of a loss line that does not contain it (:110-116: ``int("is")``), is the number of logged lines x print_every):

    get_optimizer(model, optimizer_type in {'sgd','RMSprop','Adam','lbfgs'}, learning_rate, momentum1)
    save_model(model, num_epoch, path)   -> path + "wavenet_autoencoder{N}.model"
    load_model(model, path, model_name)
    train()      reads ./params/train_params.json, ./params/model_params.json, ./params/dataset_params.json

Optional key in train_params.json (as in music_amd/train.py): ``"fused_step"`` (bool, Adam only) - the whole step runs
as forward + one softmax/CE/backward kernel + backward + flat Adam on the engine, without autograd.
"""
import glob
import os

import torch
import torch.nn as nn
import torch.optim as optim

try:
    from . import dist as wdist
    from .faster_audio_data import audio_data_loader
    from .model1 import wavenet_autoencoder
    from .train import get_params, load_model
except ImportError:
    from music_amd import dist as wdist
    from music_amd.faster_audio_data import audio_data_loader
    from music_amd.model1 import wavenet_autoencoder
    from music_amd.train import get_params, load_model

PREFIX = "wavenet_autoencoder"


def get_arguments():
    return (get_params('./params/train_params.json'), get_params('./params/model_params.json'),
            get_params('./params/dataset_params.json'))


def get_optimizer(model, optimizer_type, learning_rate, momentum1=False):
    """wavenet_autoencoder/train.py:26-34 (with ``optim.sgd`` spelled ``optim.SGD``)."""
    if optimizer_type == 'sgd':
        return optim.SGD(model.parameters(), lr=learning_rate, momentum=momentum1 or 0)
    if optimizer_type == 'RMSprop':
        return optim.RMSprop(model.parameters(), lr=learning_rate, momentum=momentum1 or 0)
    if optimizer_type == 'Adam':
        return optim.Adam(model.parameters(), lr=learning_rate)
    if optimizer_type == 'lbfgs':
        return optim.LBFGS(model.parameters(), lr=learning_rate)


def save_model(model, num_epoch, path):
    checkpoint_path = path + PREFIX + str(num_epoch) + '.model'
    print('Storing checkpoint to {}...'.format(path))
    torch.save({k: v.detach().cpu().clone() for k, v in model.state_dict().items()}, checkpoint_path)
    print('done')


def _resume_counter(log_dir, print_every):
    """Batches trained so far.  The reference reads word 2 of the last loss line (:110-116), but its own lines are
    "Average loss is X" (word 2 = "is"), so a second run on the same log directory dies there; one line is written
    every print_every batches, so the count is recovered from the number of lines (a line in the wavenet/train.py
    format "Trained over N pieces,..." - written by round-1 versions of this module - is honoured too)."""
    try:
        with open(log_dir + 'loss_log.log', 'r') as f:
            lines = [l for l in f.readlines() if l.strip()]
    except FileNotFoundError:
        return 0
    if not lines:
        return 0
    if lines[-1].startswith("Trained over "):
        return int(lines[-1].split(' ')[2])
    return len(lines) * print_every


def _epoch_of(name):
    return int(os.path.basename(name).split('.')[0][len(PREFIX):])


def train():
    cuda_available = torch.cuda.is_available()
    train_params, model_params, dataset_params = get_arguments()
    rank, world, _ = wdist.init_from_env()
    if world > 1:
        train_params = dict(train_params, seed=wdist.shared_seed(train_params.get("seed")))
    elif train_params.get("seed") is not None:
        torch.manual_seed(int(train_params["seed"]))
    net = wavenet_autoencoder(**model_params)
    epoch_trained = 0
    if train_params["restore_model"]:
        restored = load_model(net, train_params["restore_dir"], train_params["restore_model"])
        if restored is None:
            print("Initialize network and train from scratch.")
        else:
            epoch_trained = _epoch_of(train_params["restore_model"])
    if cuda_available is False and train_params["device_ids"] is not None:
        raise ValueError("Cuda is not avalable,", " can not train model using multi-gpu.")
    if world > 1:
        assert dataset_params["batch_size"] % world == 0
        dataset_params = dict(dataset_params, shard=(rank, world))
    dataloader = audio_data_loader(**dataset_params)
    if cuda_available:
        net = net.cuda()
    wdist.broadcast_parameters(list(net.parameters()))
    optimizer = get_optimizer(net, train_params.get("optimizer_type", train_params.get("optimizer", "Adam")),
                              train_params["learning_rate"], train_params.get("momentum", False))
    loss_func = nn.CrossEntropyLoss()
    is_writer = rank == 0
    loss_log_file = store_log_file = None
    if is_writer:
        os.makedirs(train_params["log_dir"], exist_ok=True)
        os.makedirs(train_params["restore_dir"], exist_ok=True)
        loss_log_file = open(train_params["log_dir"] + 'loss_log.log', 'a')
        store_log_file = open(train_params["log_dir"] + 'store_log.log', 'a')
    if world > 1:
        torch.distributed.barrier()
    num_trained = _resume_counter(train_params["log_dir"], train_params["print_every"])
    device = next(net.parameters()).device
    total_loss = torch.zeros((), dtype=torch.float64, device=device)
    step_seed = int(train_params.get("seed") or 0)
    fused = (bool(train_params.get("fused_step")) and cuda_available and
             str(train_params.get("optimizer_type", train_params.get("optimizer", "Adam"))).lower() == "adam")
    engine = None
    if fused:
        engine = net._engine_for(device)
        engine.adam_init(lr=train_params["learning_rate"])
    for epoch in range(train_params["num_epochs"]):
        for i_batch, sampled_batch in enumerate(dataloader):
            piece, target = sampled_batch["audio_piece"], sampled_batch["audio_target"]
            dp_scale = float(sampled_batch.get("dp_scale", 1.0))      # ragged last batch, see faster_audio_data._Collate
            if piece is not None:
                target = target.view(-1)
            if world > 1:
                # every replica must draw the SAME per-forward conditioning projections
                torch.manual_seed(step_seed + num_trained)

            def closure():
                optimizer.zero_grad()
                loss = torch.zeros((), device=device)
                if piece is not None:
                    loss = loss_func(net(piece), target)
                    loss.backward()
                wdist.allreduce_gradients(net.parameters(), average=True, scale=dp_scale)
                return loss
            if fused:
                loss = torch.zeros((), device=device)
                if piece is not None:
                    loss = engine.loss_and_grad(piece.to(device).float().contiguous(), target.to(device), net._draw_conditioning())
                else:
                    engine.flat_grad.zero_()
                wdist.allreduce_flat_(engine.flat_grad, average=False, scale=dp_scale)
                engine.adam_step(gscale=1.0 / wdist.world())
            else:
                loss = optimizer.step(closure) if isinstance(optimizer, optim.LBFGS) else closure()
                if not isinstance(optimizer, optim.LBFGS):
                    optimizer.step()
            loss = loss.detach() * dp_scale
            total_loss += loss.detach().double()
            num_trained += 1
            if num_trained % train_params["print_every"] == 0:
                if world > 1:
                    torch.distributed.all_reduce(total_loss)
                    total_loss /= world
                if is_writer:
                    loss_log_file.writelines('Average loss is ' + str(total_loss.item() / train_params["print_every"]) + '\n')
                    loss_log_file.flush()
                total_loss.zero_()
        if (epoch + 1) % train_params["check_point_every"] == 0 and is_writer:
            stored = glob.glob(train_params["restore_dir"] + "*.model")
            if len(stored) == train_params["max_check_points"]:
                os.remove(sorted(stored, key=_epoch_of)[0])
            save_model(net, epoch_trained + epoch + 1, train_params["restore_dir"])
            store_log_file.writelines('Epoch' + str(epoch_trained + epoch + 1) + 'model saved!')
            store_log_file.flush()
    if is_writer:
        loss_log_file.close()
        store_log_file.close()


if __name__ == '__main__':
    train()
