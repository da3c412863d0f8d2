"""query_database — list the tracking database, or one model's records (src/cae_tools/cli/query_database.py:18-28)."""
import argparse

from ..utils.model_database import ModelDatabase


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("database_path")
    parser.add_argument("--model-id", type=str, help="Dump details for this specific model", default=None)
    args = parser.parse_args(argv)
    db = ModelDatabase(args.database_path)
    if args.model_id:
        db.dump_model(model_id=args.model_id)
    else:
        db.dump()


if __name__ == "__main__":
    main()
