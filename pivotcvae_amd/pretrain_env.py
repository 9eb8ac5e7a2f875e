"""Training the click model on the HIP path: mirror of reference pretrain_env.py:25-139 (``train_response_model``).

Same arguments, same loop (shuffled DataLoader, BCE of the sigmoid of the click logits, Adam with weight decay over ALL
parameters including the item / user tables, validation each epoch, best-model pickle, three strikes of patience), with
the arithmetic on the kernels of libpcvae_hip.so: ``UserResponseModel_MLP.forward_train`` (gather -> whole-vector
normalisation -> ReLU MLP, hand-written backward incl. the embedding scatter-add), ``ops.bce_sigmoid`` and the flat-buffer
Adam.  ``ResponseTrainer`` is the step on its own (tests, callers with their own loop).
"""
import numpy as np
import torch

from . import ops
from .env.response_model import UserResponseModel_MLP
from .optim import FlatAdam


def _batch(batch_data, device):
    slates = torch.as_tensor(np.asarray(batch_data["slates"]), dtype=torch.long).to(device)
    users = torch.as_tensor(np.asarray(batch_data["users"]), dtype=torch.long).to(device)
    targets = torch.as_tensor(np.asarray(batch_data["responses"])).to(torch.float).to(device)
    return slates, users, targets


class ResponseTrainer:
    """zero_grad -> forward -> BCE(sigmoid) -> backward -> Adam(weight_decay) (pretrain_env.py:76-92)"""

    def __init__(self, model, lr, decay=0.0):
        self.model = model
        self.opt = FlatAdam(model, lr, weight_decay=decay)

    def loss(self, slates, users, targets):
        pred = self.model.forward_train(slates, users)
        return ops.bce_sigmoid(pred.reshape(-1), targets.reshape(-1))

    def step(self, slates, users, targets):
        self.opt.zero_grad()
        loss = self.loss(slates, users, targets)
        loss.backward()
        self.opt.step()
        return loss.detach()

    @torch.no_grad()
    def validation_loss(self, slates, users, targets):
        pred = self.model(slates, users)
        return ops.bce_sigmoid(pred.reshape(-1), targets.reshape(-1))


def train_response_model(trainset, valset, f_size, s_size, struct, bs, epochs, lr, decay, device, model_path, logger):
    """reference pretrain_env.py:25-139; ``trainset`` / ``valset``: datasets whose items are dicts with "slates", "users",
    "responses" and that carry ``max_iid``, ``max_uid``, ``noUser`` (data_loader.UserSlateResponseDataset)."""
    from torch.utils.data import DataLoader
    logger.log("Train user response model as simulator")
    for k, v in (("feature size", f_size), ("slate size", s_size), ("struct", struct), ("batch size", bs),
                 ("number of epoch", epochs), ("learning rate", lr)):
        logger.log("\t" + k + ": " + str(v))
    logger.log("\tdevice: " + device)
    model = UserResponseModel_MLP(trainset.max_iid, trainset.max_uid, f_size, s_size, struct, device, trainset.noUser)
    model.to(device)
    train_loader = DataLoader(trainset, batch_size=bs, shuffle=True, num_workers=0)
    val_loader = DataLoader(valset, batch_size=bs, shuffle=False, num_workers=0)
    trainer = ResponseTrainer(model, lr, decay)
    train_history, val_history = [], []
    best_val, temper = float("inf"), 3
    for epoch in range(epochs):
        logger.log("Epoch " + str(epoch + 1))
        losses = [trainer.step(*_batch(b, device)) for b in train_loader]   # device scalars: no per-step host sync
        train_history.append(float(torch.stack(losses).mean()))
        logger.log("train loss: " + str(train_history[-1]))
        vl = [trainer.validation_loss(*_batch(b, device)) for b in val_loader]
        val_history.append(float(torch.stack(vl).mean()))
        logger.log("Validation Loss: " + str(val_history[-1]))
        if epoch == 0 or val_history[-1] < best_val - 1e-4:
            torch.save(model, open(model_path, "wb"))
            logger.log("Save best model")
            temper, best_val = 3, val_history[-1]
        else:
            temper -= 1
            logger.log("Temper down to " + str(temper))
            if temper == 0:
                logger.log("Out of temper, early termination.")
                break
    return model, train_history, val_history
