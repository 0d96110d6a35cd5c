"""Entry point with the reference's shape (main.py:10-23): load params, build the mission, execute.

    python -m ippmarl.main [config.yaml] [--envs N] [--log-dir DIR]
"""
import argparse

from .missions.mission_factories import MissionFactory
from .params import load_params


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("config", nargs="?", default=None)
    ap.add_argument("--envs", type=int, default=None, help="lock-step episodes per update (default: the reference's 5 at 4 UAVs)")
    ap.add_argument("--log-dir", default="logs")
    args = ap.parse_args(argv)
    params = load_params(args.config)
    mission = MissionFactory(params, log_dir=args.log_dir, n_envs=args.envs).create_mission()
    return mission.execute()


if __name__ == "__main__":
    print(main())
