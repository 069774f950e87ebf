"""Drop-in for the reference's `server.py` entry point:  python server.py --host 0.0.0.0 --port 8020 --workers N
(the implementation lives in voice-tts_amd/server.py; `app` is what `uvicorn server:app` serves)."""
from voice_tts_amd.server import create_app, main

app = create_app()

if __name__ == "__main__":
    main()
