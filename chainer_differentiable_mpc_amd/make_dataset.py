"""Expert data set of the imitation experiment - `main(n_train, n_val, n_test)` of env_dx/make_dataset.py:16-34:
an `IL_Env('pendulum', lqr_iter=500)` populated under the true cost (seed 0) and pickled to `data/pendulum.pkl`,
which env_dx/il_exp.py:41-45 loads back.  The trajectories are computed on the GPU by `BoxDDP`; the pickle holds
them as float64 numpy arrays `[n, T, n_state + n_ctrl]` like the reference's (`IL_Env.__getstate__`)."""
import os
import pickle as pkl

from .il_env import IL_Env


def main(n_train, n_val, n_test, path=None, lqr_iter=500, device="cuda"):
    """-> the path of the pickle written (default: `<package>/data/pendulum.pkl`, make_dataset.py:23-34)"""
    if path is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "pendulum.pkl")
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    env = IL_Env('pendulum', lqr_iter=lqr_iter, device=device)
    env.populate_data(n_train=n_train, n_val=n_val, n_test=n_test, seed=0)
    with open(path, 'wb') as f:
        pkl.dump(env, f)
    return path


def load(path, device="cuda"):
    """the environment back from its pickle (il_exp.py:44-45), data moved to `device`"""
    with open(path, 'rb') as f:
        env = pkl.load(f)
    return env.to(device)
